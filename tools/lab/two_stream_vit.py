"""Lab (GPU box): does the tower gain from running frame groups on separate HIP streams?  Frames are independent through the tower and the
connector's first RegStage, and every kernel of the tower has a ramp and a tail (the last round of tiles of a persistent GEMM, the launch-to-launch
drain); with G groups on G streams the tail of one group's kernel can run under the head of another's.  Prints ms for the 26-layer tower on the
bench clip: one call of 32 frames against G in {2, 4} calls of 32 / G frames on G streams (bit-identical features are asserted).
usage: python3 tools/lab/two_stream_vit.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model = bench.build_model(dev)
    video, ids, am = bench.synthetic_inputs(dev)
    body = model.get_vision_tower().vision_tower
    nl = model.get_vision_tower()._n_layers()
    with torch.no_grad():
        ref, _ = body.encode(video, nl)
        print("one stream, 32 frames: median %.3f ms, best %.3f" % timed(lambda: body.encode(video, nl)))
        for G in (2, 4):
            streams = [torch.cuda.Stream() for _ in range(G)]
            per = video.shape[0] // G
            outs = [None] * G

            def split():
                cur = torch.cuda.current_stream()
                ev = torch.cuda.Event(); ev.record(cur)
                for g, s in enumerate(streams):
                    s.wait_event(ev)
                    with torch.cuda.stream(s):
                        outs[g], _ = body.encode(video[g * per:(g + 1) * per], nl)
                for s in streams:
                    cur.wait_stream(s)
            split(); torch.cuda.synchronize()
            got = torch.cat(outs, 0)
            print("%d streams x %d frames: median %.3f ms, best %.3f   identical: %s" % ((G, per) + timed(split) + (torch.equal(got, ref),)))


if __name__ == "__main__":
    main()
