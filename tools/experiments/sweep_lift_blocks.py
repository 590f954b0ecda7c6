import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from interactvlm_amd import ops, synth, synthetic, _lib
lib = _lib.load(); dev = torch.device("cuda:0")
V,H,W,NV,NP=4,1024,1024,6890,2048
def timeit(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/iters*1e3
vid_t,bary_t=synthetic.body_lift_tables(dev); vid_t,bary_t=vid_t.contiguous(),bary_t.contiguous()
lg=torch.randn(1,V,H,W,device=dev)*4
pid=torch.from_numpy(synth.synth_point_maps(1,V,H,W,NP,seed=0)).to(dev,torch.int32); pr=torch.rand(1,V,H,W,device=dev)
for bpc in (1, 2):
    lib.ivlm_lift_stream_blocks_per_cu(bpc)
    print(f"bpc {bpc}: dense(body) {timeit(lambda: ops.lift_mesh_dense(lg,vid_t,bary_t,NV)):.1f} us   points {timeit(lambda: ops.lift_points(pr,pid,NP)):.1f} us", flush=True)
