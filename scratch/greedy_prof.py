import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, multimodal_amd as mm
d = torch.device("cuda:0")
c = bench.CONFIGS["local_config_nocat"]
m = bench.build_model(c, d).eval()
ir, nmr, _ = bench.synth_batch(c, 1024, d, 1)
with mm.compute_dtype(torch.bfloat16):
    m.generate(nmr[:8], ir[:8], max_len=4, early_stop=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.generate(nmr, ir, max_len=80, early_stop=False)
    torch.cuda.synchronize(); print("sec", time.perf_counter() - t0)
