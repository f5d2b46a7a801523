"""two PROCESSES sharing the GPU, each running the hash-grid forward (k_grid_fwd_lean through the operator) over and over on fixed
inputs and comparing every output with its first one, bit for bit.  A kernel that is only deterministic when it has the GPU to
itself shows up here (the looping form of the lean kernel rendered runs of 16 rows differently from run to
run beside another process).  python tools/encoder_two_proc_stress.py [iterations] [rows]"""
import os, subprocess, sys

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, os.getcwd())
    from laenerf_amd.gridencoder import GridEncoder
    rank, iters, rows = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dev = "cuda:0"
    torch.manual_seed(100 + rank)
    enc = GridEncoder(desired_resolution=2048).to(dev)
    enc.embeddings.data.uniform_(-0.5, 0.5)
    # ray-ordered rows (neighbouring rows = neighbouring samples of a ray) and random ones, like frames and training batches
    t = torch.linspace(0, 1, 64, device=dev)[None, :, None]
    o = torch.rand(rows // 64, 1, 3, device=dev) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(rows // 64, 1, 3, device=dev), dim=-1)
    x_ray = (o + d * t * 0.5).clamp(-1, 1).reshape(-1, 3).contiguous()
    x_rnd = (torch.rand(rows, 3, device=dev) * 2 - 1).contiguous()
    bad = 0
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        ref = [enc(x, bound=1).clone() for x in (x_ray, x_rnd)]
        torch.cuda.synchronize()
        for it in range(iters):
            for k, x in enumerate((x_ray, x_rnd)):
                out = enc(x, bound=1)
                ne = (out.view(torch.int16) != ref[k].view(torch.int16)).any(dim=1)
                n = int(ne.sum())
                if n:
                    bad += 1
                    idx = ne.nonzero().flatten()
                    print(f"[proc {rank}] iteration {it} input {k}: {n} rows differ, first {idx[:6].tolist()}", flush=True)
    print(f"[proc {rank}] {iters} iterations x 2 inputs of {rows} rows: {bad} outputs differed from the first", flush=True)
    sys.exit(1 if bad else 0)

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
procs = [subprocess.Popen([sys.executable, __file__, "--child", str(r), str(iters), str(rows)]) for r in range(2)]
rc = [p.wait() for p in procs]
print("two-process encoder stress:", "OK" if not any(rc) else f"FAILED {rc}")
sys.exit(1 if any(rc) else 0)
