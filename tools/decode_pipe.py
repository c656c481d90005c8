"""Decode (BASELINE configs[4], beam 5, batch 128) as a pipeline (CaptionEngine.decode_pipelined): program copies x decoders in
flight, graph replay against plan walks.
    python tools/decode_pipe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from myimagecaptioningmodel_amd import default_cfg
from myimagecaptioningmodel_amd.model import CaptionEngine

B, beam, N = 128, 5, 64
cfg = default_cfg(batch_size=B, sample_count=0, **bench.WORKLOAD)
eng = CaptionEngine(cfg, device='cuda:0', use_graph=True)
image, _ = bench.synthetic_batch(B, cfg, 1234)
image_d = torch.as_tensor(image).to('cuda:0')
cases = [(g, d, k) for g in (False, True, False, True) for d, k in ((3, 2), (4, 2), (3, 3), (4, 3), (5, 3), (6, 3))]
if len(sys.argv) > 1 and sys.argv[1] == 'one':          # one case (for a kernel trace): one <copies> <decoders> <graph 0|1>
    cases = [(bool(int(sys.argv[4])), int(sys.argv[2]), int(sys.argv[3]))]
    N = 8
for use_graph, depth, decoders in cases:
    if True:
        feeds = [image_d] * N
        for _ in range(2):
            eng.decode_pipelined(feeds[:2 * depth], beam=beam, depth=depth, decoders=decoders, graph=use_graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.decode_pipelined(feeds, beam=beam, depth=depth, decoders=decoders, graph=use_graph)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('graph %-5s copies %d decoders %d: %.3f ms per batch (host returned after %.3f ms per batch), %.0f captions/s'
              % (use_graph, depth, decoders, dt / N * 1e3, th / N * 1e3, B * N / dt), flush=True)
