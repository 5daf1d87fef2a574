"""Is the DeepFM fp32 step (kernel by kernel, no graph) bound by the host issuing it?  Host issue time vs total time per step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mindrec_amd.deepfm import DeepFMConfig, DeepFMEngine
dev = torch.device("cuda:0")
for dt in ("fp32", "fp16"):
    cfg = DeepFMConfig(mlp_dtype=dt)
    eng = DeepFMEngine(cfg, dev)
    B, F = cfg.batch_size, cfg.data_field_size
    g = torch.Generator(device=dev).manual_seed(1000)
    ids = torch.randint(0, cfg.data_vocab_size, (B, F), dtype=torch.int32, device=dev, generator=g)
    wts = torch.rand((B, F), device=dev, generator=g)
    label = (torch.rand((B, 1), device=dev, generator=g) < 0.3).float()
    for _ in range(10):
        eng.train_step(ids, wts, label)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(30):
            eng.train_step(ids, wts, label)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{dt}: host issue {1e3 * (t1 - t0) / 30:.3f} ms per step, total {1e3 * (t2 - t0) / 30:.3f} ms per step", flush=True)
    del eng
