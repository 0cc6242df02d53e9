"""TEST INFRASTRUCTURE (oracle): the reference's input step, src/yolo_core.cpp:104-112 --
cv::resize(input, resized, Size(net->w, net->h)) [INTER_LINEAR], cv::cvtColor(RGB2BGR), Mat2Image
(src/visualize.cpp:26-55) -- restated with numpy integer arithmetic.

The resize lives in a third-party dependency that is absent from /root/reference and from this image: OpenCV
(the reference links whatever 4.x the host has; no version is pinned in its CMakeLists).  What is restated here is
OpenCV 4.x's generic C++ path for 8-bit INTER_LINEAR (modules/imgproc/src/resize.cpp: resizeGeneric_ with
HResizeLinear / VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>):
  fx = float((dx + 0.5) * scale_x - 0.5); sx = floor(fx); fx -= sx; left / right clamp sets fx = 0
  ialpha = cvRound((1 - fx) * 2048), cvRound(fx * 2048)   (same for rows, which are only clamped)
  S_row = src[sx] * a0 + src[sx + 1] * a1
  dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
  exact 2x shrink: OpenCV switches INTER_LINEAR to its fast INTER_AREA: (sum of the 2x2 block + 2) >> 2.
PARITY UNPINNED: no OpenCV build exists here to check this restatement against, the reference holds no frame /
expected-output pairs, and OpenCV builds with IPP or other HALs do not produce these exact bytes anyway.  The GPU
kernel (dk_image_resize_u8_to_chw) is tested bit-exact against THIS file."""
import numpy as np


def _coeffs(dn, sn, clamp_frac):
    scale = 1.0 / (float(dn) / float(sn))
    d = np.arange(dn, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_frac:
        lo = s < 0
        f[lo] = 0; s[lo] = 0
        hi = s >= sn - 1
        f[hi] = 0; s[hi] = sn - 1
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    c1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, c0, c1


def resize_u8_to_chw(frames, w, h, swap_rb=False):
    """frames: uint8 [batch, sh, sw, c] -> float32 [batch, c, h, w] in [0, 1]."""
    frames = np.asarray(frames, np.uint8)
    B, sh, sw, c = frames.shape
    src = frames.astype(np.int64)
    if sw == 2 * w and sh == 2 * h:
        out = (src[:, 0::2, 0::2] + src[:, 0::2, 1::2] + src[:, 1::2, 0::2] + src[:, 1::2, 1::2] + 2) >> 2
    else:
        sx, a0, a1 = _coeffs(w, sw, True)
        sy, b0, b1 = _coeffs(h, sh, False)
        sx1 = np.minimum(sx + 1, sw - 1)
        y0 = np.clip(sy, 0, sh - 1)
        y1 = np.clip(sy + 1, 0, sh - 1)
        S = src[:, :, sx, :] * a0[None, None, :, None] + src[:, :, sx1, :] * a1[None, None, :, None]   # [B, sh, w, c]
        S0, S1 = S[:, y0], S[:, y1]
        out = (((b0[None, :, None, None] * (S0 >> 4)) >> 16) + ((b1[None, :, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
        out = np.clip(out, 0, 255)
    if swap_rb and c >= 3:
        out = out[..., [2, 1, 0] + list(range(3, c))]
    return (out.astype(np.float32) / np.float32(255.0)).transpose(0, 3, 1, 2).copy()
