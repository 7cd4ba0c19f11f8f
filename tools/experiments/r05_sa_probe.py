"""Where does runtime.SamplerAheadPipeline lose against one graph per batch?  Times, over the same captured graphs:
  a) the sampler graphs alone, back to back on nS streams          (the sampler-bound rate)
  b) the dense graphs alone on nD streams, no events               (the dense-bound rate)
  c) both kinds at their own pace, no events between them          (interference only)
  d) the pipeline itself (events)
    gpurun -- 'python tools/sa_probe.py'
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pn2_amd as pn2  # noqa: E402

dev = torch.device("cuda:0")
hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
B, N, SLOTS, K = 16, 8192, 8, 40
tfu = pn2.util.tf_util
tfu.set_default_store(tfu.VariableStore(device=dev, seed=0))
batches = [torch.from_numpy(bench.s_scene(2000 + i, B, N)).to(dev) for i in range(SLOTS)]
with torch.no_grad():
    pn2.model.get_sa_fp_features(batches[0], False, hp)


def timeit(fn, k=K):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


for nS, nD in ((1, 3), (2, 2), (1, 2), (2, 3)):
    pipe = pn2.runtime.SamplerAheadPipeline(lambda x: pn2.model.sa1_samples(x, hp),
                                            lambda x, s: pn2.model.get_sa_fp_features(x, False, hp, sa1=s)[0],
                                            batches, sampler_streams=nS, dense_streams=nD)
    c = [0]

    def samplers():
        k = c[0] % SLOTS; c[0] += 1
        with torch.cuda.stream(pipe.s_streams[k % nS]):
            pipe.s_graphs[k].replay()

    def dense():
        k = c[0] % SLOTS; c[0] += 1
        with torch.cuda.stream(pipe.d_streams[k % nD]):
            pipe.d_graphs[k].replay()

    def both():
        k = c[0] % SLOTS; c[0] += 1
        with torch.cuda.stream(pipe.s_streams[k % nS]):
            pipe.s_graphs[k].replay()
        with torch.cuda.stream(pipe.d_streams[k % nD]):
            pipe.d_graphs[k].replay()

    print("nS=%d nD=%d: samplers alone %.4f  dense alone %.4f  both, no events %.4f  pipeline %.4f  (ms per batch)"
          % (nS, nD, timeit(samplers), timeit(dense), timeit(both), timeit(pipe.step)), flush=True)
    del pipe
# one graph per batch for comparison, on 1..4 streams
fwd = lambda x: pn2.model.get_sa_fp_features(x, False, hp)[0]  # noqa: E731
caps = [pn2.runtime.CapturedForward(fwd, b_) for b_ in batches[:4]]
for P in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(P)]
    c = [0]

    def step():
        i = c[0] % P; c[0] += 1
        with torch.cuda.stream(streams[i]):
            caps[i].replay()
    print("one graph per batch on %d streams: %.4f" % (P, timeit(step)), flush=True)
