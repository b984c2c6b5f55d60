"""Where does the N>1 result exchange spend its time? (single rank, RCCL world of 1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import rustfst_amd
from rustfst_amd import synth, dist as wdist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
t = synth.make_transducer(200_000, 10, 256, 0.0, seed=3)
accs = synth.make_acceptors(t, 64, 200, seed0=1000)
ctx = rustfst_amd.default_context()
dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
outs, _ = rustfst_amd.compose_shortest_path_batch(rustfst_amd.DeviceFst.upload_many(accs, ctx), dt)
N = 50
for name, fn in (("pack_device_paths", lambda: wdist.pack_device_paths(outs, 208)),):
    for _ in range(5): fn()
    a = time.perf_counter()
    for _ in range(N): r = fn()
    print(name, "%.1f us" % ((time.perf_counter() - a) / N * 1e6))
packed = wdist.pack_device_paths(outs, 208)
for _ in range(5): wdist.gather_paths(packed, 1, dev)
torch.cuda.synchronize()
a = time.perf_counter()
for _ in range(N): g = wdist.gather_paths(packed, 1, dev)
print("gather_paths %.1f us (%d bytes)" % ((time.perf_counter() - a) / N * 1e6, packed.nbytes))
# pieces
tt = torch.from_numpy(packed.view(np.int32))
a = time.perf_counter()
for _ in range(N): d = tt.to(dev, non_blocking=True)
torch.cuda.synchronize(); print("  H2D %.1f us" % ((time.perf_counter() - a) / N * 1e6))
out = torch.empty_like(d)
a = time.perf_counter()
for _ in range(N): dist.all_gather_into_tensor(out, d)
torch.cuda.synchronize(); print("  all_gather %.1f us" % ((time.perf_counter() - a) / N * 1e6))
a = time.perf_counter()
for _ in range(N): h = out.cpu()
print("  D2H %.1f us" % ((time.perf_counter() - a) / N * 1e6))
dist.destroy_process_group()
