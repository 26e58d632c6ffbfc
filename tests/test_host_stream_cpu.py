"""CPU: the call sequence of SingleGpuPipeline.process_frame_host (frames that start in host memory, staged two ahead of their use) against a
recording stand-in for the context -- which frame is uploaded when, which staged frame becomes current, whose device address the prefetch gets,
what happens at the end of the stream and after a seek.  (The GPU side of the same path: tests/test_gpu_parity.py::test_host_stream_staged_two_ahead_...)"""
import numpy as np

from hybkinectfu_amd import pipeline as PL


class RecordingCtx:
    """mirrors the staging rules of include/hybkf.h: kf_upload_depth_mm drops what is staged, at most two frames staged, take needs one"""
    def __init__(self):
        self.calls = []
        self.staged = []
        self.current = None
        self.addr = 1000

    def upload_depth_mm(self, mm):
        self.staged = []
        self.current = int(mm[0, 0])
        self.calls.append(("upload", self.current))

    def upload_depth_mm_next(self, mm):
        assert len(self.staged) < 2, "a third staged frame is KF_ERR_STATE"
        self.addr += 1
        self.staged.append((int(mm[0, 0]), self.addr))
        self.calls.append(("upload_next", int(mm[0, 0])))
        return self.addr

    def take_next_depth(self):
        assert self.staged, "nothing staged is KF_ERR_STATE"
        self.current, _ = self.staged.pop(0)
        self.calls.append(("take", self.current))

    def preprocess(self, *a):
        self.calls.append(("preprocess", self.current))

    def prefetch_frame(self, dev, *a):
        frame = [f for f, ad in self.staged if ad == dev]
        assert len(frame) == 1, "the prefetch must name a staged frame's device address"
        self.calls.append(("prefetch", frame[0]))

    def icp_track(self, frame_id, *a):
        self.calls.append(("track", frame_id, self.current))

    def integrate(self, *a, **k):
        pass

    def raycast(self, *a, **k):
        pass


def make_pipe():
    pipe = PL.SingleGpuPipeline.__new__(PL.SingleGpuPipeline)      # (no context: the stand-in takes its place)
    pipe.ctx = RecordingCtx()
    pipe.trunc_max, pipe.integ_dist, pipe.tracker, pipe.color, pipe.inc, pipe._host = 4.0, 2.0, "icp", False, 0.1, None
    return pipe


def frame_source(n):
    return lambda k: np.full((2, 2), k, np.uint16) if k < n else None


def test_frames_are_staged_two_ahead_and_tracked_in_order():
    n = 6
    pipe = make_pipe()
    for k in range(n):
        pipe.process_frame_host(frame_source(n), k)
    c = pipe.ctx.calls
    assert [x for x in c if x[0] == "track"] == [("track", k, k) for k in range(n)]                      # frame k is the current frame when it is tracked
    assert [x[1] for x in c if x[0] in ("upload", "upload_next")] == list(range(n))                      # every frame crosses PCIe exactly once, in order
    assert [x[1] for x in c if x[0] == "prefetch"] == list(range(1, n))                                  # frame k + 1's front end rides in frame k
    assert c[:3] == [("upload", 0), ("upload_next", 1), ("preprocess", 0)]
    # steady state: take k, preprocess k, upload k + 2, prefetch k + 1, track k
    i = c.index(("take", 2))
    assert c[i:i + 5] == [("take", 2), ("preprocess", 2), ("upload_next", 4), ("prefetch", 3), ("track", 2, 2)]
    assert pipe.ctx.staged == []                                                                         # nothing left behind at the end of the stream


def test_a_seek_restarts_the_staging():
    n = 10
    pipe = make_pipe()
    for k in (0, 1, 2, 7, 8):
        pipe.process_frame_host(frame_source(n), k)
    c = pipe.ctx.calls
    assert [x for x in c if x[0] == "track"] == [("track", k, k) for k in (0, 1, 2, 7, 8)]
    i = c.index(("upload", 7))                                                                           # the seek: frame 7 is uploaded and waited for, 8 staged behind it
    assert c[i:i + 2] == [("upload", 7), ("upload_next", 8)]
    assert ("take", 8) in c and ("upload", 8) not in c


def test_a_one_frame_stream():
    pipe = make_pipe()
    pipe.process_frame_host(frame_source(1), 0)
    assert pipe.ctx.calls == [("upload", 0), ("preprocess", 0), ("track", 0, 0)]
