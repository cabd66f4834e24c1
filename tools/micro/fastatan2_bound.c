// Exhaustive bound of cv::fastAtan2 polynomial against atan over every float quotient c in [2^-31, 1] (the margin of the growth agent cheap alignment test,
// host_tables.cpp alignTanLo / alignTanHi): gcc -O2 -fopenmp -ffp-contract=off -o fastatan2_bound fastatan2_bound.c -lm && ./fastatan2_bound
// -> max err 0.009546480 deg at c=0.99999994  (= 1.666e-04 rad)
#include <stdio.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <omp.h>
int main(){
  const float k=(float)(180/3.1415926535897932384626433832795);
  const float p1=0.9997878412794807f*k,p3=-0.3258083974640975f*k,p5=0.1555786518463281f*k,p7=-0.04432655554792128f*k;
  double gmax=0; float gc=0;
  #pragma omp parallel
  { double mx=0; float mc=0;
    #pragma omp for schedule(static)
    for(long long u=0x30000000LL; u<=0x3f800000LL; ++u){ uint32_t b=(uint32_t)u; float c; memcpy(&c,&b,4);
      volatile float c2=c*c; volatile float t=p7*c2; t=t+p5; t=t*c2; t=t+p3; t=t*c2; t=t+p1; t=t*c;
      double e=fabs((double)t-atan((double)c)*(180/3.1415926535897932384626433832795));
      if(e>mx){mx=e;mc=c;} }
    #pragma omp critical
    if(mx>gmax){gmax=mx;gc=mc;} }
  printf("max err %.9f deg at c=%.9g  (= %.3e rad)\n",gmax,gc,gmax*3.14159265358979/180);
}
