#!/usr/bin/env python3
"""Author the .cfg files for the BASELINE.json configs.

The reference ships no cfg files (its .gitignore excludes cfg/), so the three
network descriptions used by tests and bench.py are generated here from the
public YOLOv4 architecture grammar, restricted to the
keys the reference parser knows (src/parser.cpp:921-1055, 179-242, 312-415).
Validation invariants (checked in tests/test_cfg.py and, in the build
container, against the compiled reference): layer count, total BFLOPS and the
byte size of the .weights file each cfg consumes.
"""
import os
import sys

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cfg")


class Cfg:
    def __init__(self):
        self.lines = []
        self.n = 0  # number of layers emitted so far (excludes [net])

    def net(self, w, h, batch=64, subdivisions=8, **kw):
        d = dict(batch=batch, subdivisions=subdivisions, width=w, height=h,
                 channels=3, momentum=0.949, decay=0.0005, angle=0,
                 saturation=1.5, exposure=1.5, hue=.1, learning_rate=0.00261,
                 burn_in=1000, max_epoch=300, policy="steps", steps=".8,.9",
                 scales=".1,.1")
        d.update(kw)
        self.lines.append("[net]")
        for k, v in d.items():
            self.lines.append(f"{k}={v}")
        self.lines.append("")

    def _sec(self, name, **kw):
        self.lines.append(f"# layer {self.n}")
        self.lines.append(f"[{name}]")
        for k, v in kw.items():
            self.lines.append(f"{k}={v}")
        self.lines.append("")
        self.n += 1
        return self.n - 1

    def conv(self, f, k, s, act, bn=1):
        kw = {}
        if bn:
            kw["batch_normalize"] = 1
        kw.update(filters=f, size=k, stride=s, pad=1, activation=act)
        return self._sec("convolutional", **kw)

    def route(self, *layers, groups=None, group_id=None):
        kw = dict(layers=",".join(str(x) for x in layers))
        if groups is not None:
            kw.update(groups=groups, group_id=group_id)
        return self._sec("route", **kw)

    def shortcut(self, frm=-3):
        return self._sec("shortcut", **{"from": frm, "activation": "linear"})

    def maxpool(self, size, stride):
        return self._sec("maxpool", size=size, stride=stride)

    def upsample(self, stride=2):
        return self._sec("upsample", stride=stride)

    def yolo(self, mask, anchors, num, **kw):
        d = dict(mask=",".join(str(m) for m in mask), anchors=anchors,
                 classes=80, num=num)
        d.update(kw)
        return self._sec("yolo", **d)

    def text(self):
        return "\n".join(self.lines) + "\n"


def csp_stage(c, f, n, act):
    """S(f,n) of SURVEY.md Appendix A.2: downsample + CSP block with n residual units."""
    c.conv(2 * f, 3, 2, act)
    c.conv(f, 1, 1, act)
    c.route(-2)
    c.conv(f, 1, 1, act)
    for _ in range(n):
        c.conv(f, 1, 1, act)
        c.conv(f, 3, 1, act)
        c.shortcut(-3)
    c.conv(f, 1, 1, act)
    c.route(-1, -(3 * n + 4))
    c.conv(2 * f, 1, 1, act)


def yolov4_tiny():
    c = Cfg()
    c.net(416, 416, batch=64, subdivisions=1, momentum=0.9)
    A = "leaky"
    c.conv(32, 3, 2, A)
    c.conv(64, 3, 2, A)
    for f in (64, 128, 256):
        c.conv(f, 3, 1, A)
        c.route(-1, groups=2, group_id=1)
        c.conv(f // 2, 3, 1, A)
        c.conv(f // 2, 3, 1, A)
        c.route(-1, -2)
        c.conv(f, 1, 1, A)
        c.route(-6, -1)
        c.maxpool(2, 2)
    c.conv(512, 3, 1, A)
    c.conv(256, 1, 1, A)
    c.conv(512, 3, 1, A)
    c.conv(255, 1, 1, "linear", bn=0)
    anchors = "10,14, 23,27, 37,58, 81,82, 135,169, 344,319"
    ykw = dict(jitter=.3, scale_x_y=1.05, cls_normalizer=1.0,
               iou_normalizer=0.07, iou_loss="ciou", ignore_thresh=.7,
               truth_thresh=1, random=0, nms_kind="greedynms", beta_nms=0.6)
    c.yolo((3, 4, 5), anchors, 6, **ykw)
    c.route(-4)
    c.conv(128, 1, 1, A)
    c.upsample(2)
    c.route(-1, 23)
    c.conv(256, 3, 1, A)
    c.conv(255, 1, 1, "linear", bn=0)
    c.yolo((1, 2, 3), anchors, 6, **ykw)
    assert c.n == 38, c.n
    return c.text()


V4_ANCHORS = "12,16, 19,36, 40,28, 36,75, 76,55, 72,146, 142,110, 192,243, 459,401"


def yolov4():
    c = Cfg()
    c.net(608, 608)
    M, L = "mish", "leaky"
    c.conv(32, 3, 1, M)
    # stage 1 (different inner widths from the generic S(f,n))
    c.conv(64, 3, 2, M)
    c.conv(64, 1, 1, M)
    c.route(-2)
    c.conv(64, 1, 1, M)
    c.conv(32, 1, 1, M)
    c.conv(64, 3, 1, M)
    c.shortcut(-3)
    c.conv(64, 1, 1, M)
    c.route(-1, -7)
    c.conv(64, 1, 1, M)
    for f, n in ((64, 2), (128, 8), (256, 8), (512, 4)):
        csp_stage(c, f, n, M)
    assert c.n == 105, c.n  # stage outputs are layers 23, 54, 85, 104
    # neck: SPP
    c.conv(512, 1, 1, L)
    c.conv(1024, 3, 1, L)
    c.conv(512, 1, 1, L)
    c.maxpool(5, 1)
    c.route(-2)
    c.maxpool(9, 1)
    c.route(-4)
    c.maxpool(13, 1)
    c.route(-1, -3, -5, -6)
    c.conv(512, 1, 1, L)
    c.conv(1024, 3, 1, L)
    c.conv(512, 1, 1, L)
    # PAN top-down
    for f, lat in ((256, 85), (128, 54)):
        c.conv(f, 1, 1, L)
        c.upsample(2)
        c.route(lat)
        c.conv(f, 1, 1, L)
        c.route(-1, -3)
        c.conv(f, 1, 1, L)
        c.conv(2 * f, 3, 1, L)
        c.conv(f, 1, 1, L)
        c.conv(2 * f, 3, 1, L)
        c.conv(f, 1, 1, L)
    ykw = dict(jitter=.3, ignore_thresh=.7, truth_thresh=1, iou_thresh=0.213,
               cls_normalizer=1.0, iou_normalizer=0.07, iou_loss="ciou",
               nms_kind="greedynms", beta_nms=0.6, max_delta=5)
    # head 1
    c.conv(256, 3, 1, L)
    c.conv(255, 1, 1, "linear", bn=0)
    c.yolo((0, 1, 2), V4_ANCHORS, 9, scale_x_y=1.2, **ykw)
    # bottom-up 1
    c.route(-4)
    c.conv(256, 3, 2, L)
    c.route(-1, -16)
    c.conv(256, 1, 1, L)
    c.conv(512, 3, 1, L)
    c.conv(256, 1, 1, L)
    c.conv(512, 3, 1, L)
    c.conv(256, 1, 1, L)
    c.conv(512, 3, 1, L)
    c.conv(255, 1, 1, "linear", bn=0)
    c.yolo((3, 4, 5), V4_ANCHORS, 9, scale_x_y=1.1, **ykw)
    # bottom-up 2
    c.route(-4)
    c.conv(512, 3, 2, L)
    c.route(-1, -37)
    c.conv(512, 1, 1, L)
    c.conv(1024, 3, 1, L)
    c.conv(512, 1, 1, L)
    c.conv(1024, 3, 1, L)
    c.conv(512, 1, 1, L)
    c.conv(1024, 3, 1, L)
    c.conv(255, 1, 1, "linear", bn=0)
    c.yolo((6, 7, 8), V4_ANCHORS, 9, scale_x_y=1.05, **ykw)
    assert c.n == 162, c.n
    return c.text()


def yolov4_csp():
    """scaled-YOLOv4 (yolov4-csp) at 512x512, SURVEY.md Appendix A.3."""
    c = Cfg()
    c.net(512, 512)
    M = "mish"
    c.conv(32, 3, 1, M)
    c.conv(64, 3, 2, M)
    c.conv(32, 1, 1, M)
    c.conv(64, 3, 1, M)
    c.shortcut(-3)
    for f, n in ((64, 2), (128, 8), (256, 8), (512, 4)):
        csp_stage(c, f, n, M)
    assert c.n == 99, c.n  # stage outputs at 48, 79, 98
    # CSP-SPP
    c.conv(512, 1, 1, M)
    c.route(-2)
    c.conv(512, 1, 1, M)
    c.conv(512, 3, 1, M)
    c.conv(512, 1, 1, M)
    c.maxpool(5, 1)
    c.route(-2)
    c.maxpool(9, 1)
    c.route(-4)
    c.maxpool(13, 1)
    c.route(-1, -3, -5, -6)
    c.conv(512, 1, 1, M)
    c.conv(512, 3, 1, M)
    c.route(-1, -13)
    c.conv(512, 1, 1, M)
    assert c.n == 114, c.n  # layer 113 is the SPP output
    for f, lat in ((256, 79), (128, 48)):
        c.conv(f, 1, 1, M)
        c.upsample(2)
        c.route(lat)
        c.conv(f, 1, 1, M)
        c.route(-1, -3)
        c.conv(f, 1, 1, M)
        c.conv(f, 1, 1, M)
        c.route(-2)
        c.conv(f, 1, 1, M)
        c.conv(f, 3, 1, M)
        c.conv(f, 1, 1, M)
        c.conv(f, 3, 1, M)
        c.route(-1, -6)
        c.conv(f, 1, 1, M)
    assert c.n == 142, c.n  # up-blocks end at 127 and 141
    ykw = dict(jitter=.1, scale_x_y=2.0, ignore_thresh=.7, truth_thresh=1,
               iou_thresh=0.2, cls_normalizer=0.5, iou_normalizer=0.05,
               iou_loss="ciou", nms_kind="diounms", beta_nms=0.6, max_delta=2)

    def down(f, back):
        c.route(-4)
        c.conv(f, 3, 2, M)
        c.route(-1, back)
        c.conv(f, 1, 1, M)
        c.conv(f, 1, 1, M)
        c.route(-2)
        c.conv(f, 1, 1, M)
        c.conv(f, 3, 1, M)
        c.conv(f, 1, 1, M)
        c.conv(f, 3, 1, M)
        c.route(-1, -6)
        c.conv(f, 1, 1, M)

    c.conv(256, 3, 1, M)
    c.conv(255, 1, 1, "logistic", bn=0)
    c.yolo((0, 1, 2), V4_ANCHORS, 9, **ykw)
    down(256, -20)
    c.conv(512, 3, 1, M)
    c.conv(255, 1, 1, "logistic", bn=0)
    c.yolo((3, 4, 5), V4_ANCHORS, 9, **ykw)
    down(512, -49)
    c.conv(1024, 3, 1, M)
    c.conv(255, 1, 1, "logistic", bn=0)
    c.yolo((6, 7, 8), V4_ANCHORS, 9, **ykw)
    assert c.n == 175, c.n
    return c.text()


def yolov4x_mish(size=512):
    """scaled-YOLOv4 "x" (yolov4x-mish, BASELINE configs[4]): the yolov4-csp block grammar at width
    x1.25 and depth x1.33 -- stem 32 / 80, CSP stages (half width, residual units) (80,3) (160,10)
    (320,10) (640,5), CSP-SPP at 640, PAN blocks with THREE (1x1, 3x3) pairs, heads 320 / 640 / 1280.
    Written from the public architecture description; there is no network access to diff it against
    the upstream file, so the invariants printed by tools/make_golden.py cfgcheck (layer count,
    BFLOPS, weight bytes, from the reference's own parser) are this repo's record, not upstream's."""
    c = Cfg()
    c.net(size, size)
    M = "mish"
    c.conv(32, 3, 1, M)
    c.conv(80, 3, 2, M)
    c.conv(40, 1, 1, M)
    c.conv(80, 3, 1, M)
    c.shortcut(-3)
    stage_out = []
    for f, n in ((80, 3), (160, 10), (320, 10), (640, 5)):
        csp_stage(c, f, n, M)
        stage_out.append(c.n - 1)
    p3, p4 = stage_out[1], stage_out[2]   # stride-8 and stride-16 backbone outputs
    # CSP-SPP
    c.conv(640, 1, 1, M)
    c.route(-2)
    c.conv(640, 1, 1, M)
    c.conv(640, 3, 1, M)
    c.conv(640, 1, 1, M)
    c.maxpool(5, 1)
    c.route(-2)
    c.maxpool(9, 1)
    c.route(-4)
    c.maxpool(13, 1)
    c.route(-1, -3, -5, -6)
    c.conv(640, 1, 1, M)
    c.conv(640, 3, 1, M)
    c.route(-1, -13)
    spp = c.conv(640, 1, 1, M)
    PAIRS = 3

    def csp_neck(f):
        """[1x1 f] [route -2] [1x1 f] PAIRS x ([3x3 f] ... ) [route] [1x1 f]"""
        c.conv(f, 1, 1, M)
        c.route(-2)
        c.conv(f, 1, 1, M)
        for k in range(PAIRS):
            c.conv(f, 3, 1, M)
            if k + 1 < PAIRS:
                c.conv(f, 1, 1, M)
        c.route(-1, -(2 * PAIRS + 2))
        return c.conv(f, 1, 1, M)

    ups = []
    for f, lat in ((320, p4), (160, p3)):
        c.conv(f, 1, 1, M)
        c.upsample(2)
        c.route(lat)
        c.conv(f, 1, 1, M)
        c.route(-1, -3)
        c.conv(f, 1, 1, M)
        ups.append(csp_neck(f))
    ykw = dict(jitter=.1, scale_x_y=2.0, ignore_thresh=.7, truth_thresh=1,
               iou_thresh=0.2, cls_normalizer=0.5, iou_normalizer=0.05,
               iou_loss="ciou", nms_kind="diounms", beta_nms=0.6, max_delta=2)

    def head(f, mask):
        c.conv(f, 3, 1, M)
        c.conv(255, 1, 1, "logistic", bn=0)
        c.yolo(mask, V4_ANCHORS, 9, **ykw)

    def down(f, lateral):
        c.route(-4)               # the neck block output below the head
        c.conv(f, 3, 2, M)
        c.route(-1, lateral)      # absolute index of the same-resolution top-down block
        c.conv(f, 1, 1, M)
        return csp_neck(f)

    head(320, (0, 1, 2))
    down(320, ups[0])
    head(640, (3, 4, 5))
    down(640, spp)
    head(1280, (6, 7, 8))
    return c.text()


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, fn in (("yolov4-tiny.cfg", yolov4_tiny), ("yolov4.cfg", yolov4),
                     ("yolov4-csp.cfg", yolov4_csp), ("yolov4x-mish.cfg", yolov4x_mish)):
        with open(os.path.join(OUT, name), "w") as f:
            f.write(fn())
        print("wrote", name)


if __name__ == "__main__":
    sys.exit(main())
