"""bench.py's input options on the GPU (the default synthetic run is what the driver times; this covers --images)."""
import json
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_on_a_recorded_sequence_directory(tmp_path):
    """bench.py --images on a KITTI-shaped directory (times.txt, image_0/%06d.png, image_1/%06d.png -- Examples/PL/PL_stereo_kitti.cc:130-160 LoadImages): the image
    size comes from the files (here 600 x 368, none of the named configurations), the feature counts from --config, the JSON line is the synthetic run's."""
    import numpy as np
    from PIL import Image
    from orb_line_slam_amd import synth
    w, h, n = 600, 368, 6
    imgs = synth.stereo_batch(123, n, w, h)
    k = tmp_path / "kitti" / "07"
    (k / "image_0").mkdir(parents=True); (k / "image_1").mkdir()
    (k / "times.txt").write_text("".join("%e\n" % (0.1 * i) for i in range(n)))
    for i in range(n):
        Image.fromarray(imgs[2 * i]).save(k / "image_0" / ("%06d.png" % i))
        Image.fromarray(imgs[2 * i + 1]).save(k / "image_1" / ("%06d.png" % i))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--images", str(k), "--config", "C2", "--pairs", "8", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-extras", "--no-isolated"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    d = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith("{")][0])
    assert d["data"] == "recorded: 07" and d["value"] > 0 and d["n_gpus"] == 1
    assert d["config"]["pairs_per_gpu_per_step"] == 8 and d["config"]["mean_keylines_per_image"] > 20
