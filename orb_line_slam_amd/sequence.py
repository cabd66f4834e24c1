"""Recorded stereo sequences as input of the batched offline mode (BASELINE configs C1 / C4): the file lists the reference's example drivers
build -- KITTI odometry layout, `LoadImages` of Examples/PL/PL_stereo_kitti.cc:130-160 (`<seq>/times.txt`, `<seq>/image_0/%06d.png`,
`<seq>/image_1/%06d.png`), and the EuRoC layout, `LoadImages` of Examples/PL/PL_stereo_euroc.cc:192-216 (one stamp per line of the times file,
`<left>/<stamp>.png`, `<right>/<stamp>.png`, time = stamp / 1e9) -- and a reader that decodes them into the (2 * pairs, H, W) uint8 batches
`OfflinePipeline.run` / `bench.py --images` take (left image of pair p at 2p, right at 2p + 1, as everywhere in this package).

Decoding is host work (PIL: PNG, PGM, JPEG ...).  8-bit grey files are passed through unchanged, which is what `cv::imread(..., CV_LOAD_IMAGE_UNCHANGED)`
hands to `System::TrackStereo`; colour files are handed over in imread's B, G, R order and converted by this package's own `cvtColor` with the code
`Tracking::GrabImageStereo` (src/Tracking.cc:193-218) picks from `Camera.RGB` -- see `read_gray` -- never with PIL's luma formula, whose rounding differs.
"""
import os
import numpy as np


def load_images_kitti(path_to_sequence):
    """(left files, right files, timestamps) of a KITTI odometry sequence directory -- Examples/PL/PL_stereo_kitti.cc:130-160."""
    times = []
    with open(os.path.join(path_to_sequence, "times.txt")) as f:
        for line in f:
            s = line.strip()
            if s:
                times.append(float(s.split()[0]))
    left = [os.path.join(path_to_sequence, "image_0", "%06d.png" % i) for i in range(len(times))]
    right = [os.path.join(path_to_sequence, "image_1", "%06d.png" % i) for i in range(len(times))]
    return left, right, times


def load_images_euroc(path_left, path_right, path_times):
    """(left files, right files, timestamps [s]) of a EuRoC recording -- Examples/PL/PL_stereo_euroc.cc:192-216."""
    left, right, times = [], [], []
    with open(path_times) as f:
        for line in f:
            s = line.strip()
            if s:
                left.append(os.path.join(path_left, s + ".png"))
                right.append(os.path.join(path_right, s + ".png"))
                times.append(float(s.split()[0]) / 1e9)
    return left, right, times


def find_sequence(path):
    """Accepts what a user is likely to point at: a KITTI sequence directory (times.txt + image_0 + image_1), a EuRoC `mav0` directory
    (cam0/data, cam1/data; stamps from cam0/data.csv or the sorted file names), or any directory with two sub-directories of equally many image
    files (`left` / `right`, `image_0` / `image_1`, `cam0` / `cam1`; pairs are matched in sorted order).  Returns (left files, right files, times)."""
    if os.path.isfile(os.path.join(path, "times.txt")) and os.path.isdir(os.path.join(path, "image_0")):
        return load_images_kitti(path)
    exts = (".png", ".pgm", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff", ".ppm")

    def listing(d):
        return sorted(os.path.join(d, f) for f in os.listdir(d) if f.lower().endswith(exts))
    for l, r in (("cam0/data", "cam1/data"), ("left", "right"), ("image_0", "image_1"), ("cam0", "cam1"), ("image_2", "image_3")):
        dl, dr = os.path.join(path, l), os.path.join(path, r)
        if os.path.isdir(dl) and os.path.isdir(dr):
            fl, fr = listing(dl), listing(dr)
            if len(fl) != len(fr):
                raise ValueError("ERROR: Different number of left and right images.")      # PL_stereo_euroc.cc:59
            if not fl:
                continue
            times = []
            for f in fl:
                stem = os.path.splitext(os.path.basename(f))[0]
                times.append(float(stem) / 1e9 if stem.isdigit() and len(stem) > 12 else float(len(times)))
            return fl, fr, times
    raise FileNotFoundError(f"{path}: neither a KITTI sequence (times.txt, image_0, image_1) nor a directory with left / right image folders")


def read_gray(path, camera_rgb=True, context=None):
    """One image file as the (H, W) uint8 image the reference's tracker works on (src/Tracking.cc:193-218).  Grey files pass unchanged.  A colour file is what
    `cv::imread(..., CV_LOAD_IMAGE_UNCHANGED)` delivers -- channels in B, G, R(, A) order -- put through `cvtColor` with the code `Tracking::GrabImageStereo`
    picks from `Camera.RGB` (`camera_rgb`): CV_RGB2GRAY / CV_RGBA2GRAY when it is 1 -- which every KITTI / EuRoC yaml of the reference sets
    (Examples/PL/PL_KITTI00-02.yaml:32), so the reference weighs imread's BLUE channel with 0.299 and its RED channel with 0.114 -- and CV_BGR2GRAY / CV_BGRA2GRAY
    when it is 0.  The conversion is this package's own kernel (`precond.cvtColor`: needs the GPU), never PIL's luma formula, whose rounding differs."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode == "L":
            return np.asarray(im, dtype=np.uint8)
        if im.mode in ("I;16", "I;16B", "I;16L", "I", "F"):
            raise ValueError(f"{path}: 16-bit / float images are not a supported input (the reference's pipeline is 8-bit)")
        alpha = im.mode in ("RGBA", "LA", "PA") or (im.mode == "P" and "transparency" in im.info)
        px = np.asarray(im.convert("RGBA" if alpha else "RGB"), dtype=np.uint8)
    from .precond import cvtColor, RGB2GRAY, BGR2GRAY, RGBA2GRAY, BGRA2GRAY
    # imread's channel order: B, G, R (, A)
    bgr = np.ascontiguousarray(np.concatenate([px[..., 2::-1], px[..., 3:]], axis=-1))
    code = (RGBA2GRAY if camera_rgb else BGRA2GRAY) if alpha else (RGB2GRAY if camera_rgb else BGR2GRAY)
    return cvtColor(bgr[None], code, context=context)[0]


class StereoSequence:
    """Batches of a recorded stereo sequence: `for batch, times in StereoSequence(path).batches(pairs)` yields (2 * n, H, W) uint8 arrays, n <= pairs.
    `out` = a function i -> writable (2 * pairs, H, W) array (e.g. OfflinePipeline.input_buffer) decodes straight into pinned staging memory."""

    def __init__(self, path=None, left=None, right=None, times=None, limit=None, camera_rgb=True):
        # camera_rgb: the yaml's Camera.RGB (1 in every KITTI / EuRoC configuration of the reference); only matters for colour files (read_gray)
        self.camera_rgb = bool(camera_rgb)
        if path is not None:
            left, right, times = find_sequence(path)
        if left is None or right is None or len(left) != len(right):
            raise ValueError("ERROR: Different number of left and right images.")
        if limit:
            left, right, times = left[:limit], right[:limit], (times[:limit] if times is not None else None)
        self.left, self.right = list(left), list(right)
        self.times = list(times) if times is not None else [float(i) for i in range(len(self.left))]
        first = read_gray(self.left[0], self.camera_rgb)
        self.height, self.width = first.shape

    def __len__(self):
        return len(self.left)

    def batches(self, pairs, out=None):
        for b, start in enumerate(range(0, len(self.left), pairs)):
            n = min(pairs, len(self.left) - start)
            buf = out(b) if out is not None else np.empty((2 * pairs, self.height, self.width), np.uint8)
            for k in range(n):
                for side, files in ((0, self.left), (1, self.right)):
                    img = read_gray(files[start + k], self.camera_rgb)
                    if img.shape != (self.height, self.width):
                        raise ValueError(f"{files[start + k]}: {img.shape[1]}x{img.shape[0]}, the sequence started with {self.width}x{self.height}")
                    buf[2 * k + side] = img
            yield buf[:2 * n], self.times[start:start + n]
