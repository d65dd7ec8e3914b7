"""GPU soak (not collected by pytest): N extra fuzz seeds through the steady-state entry point against the oracle,
every third seed on a context with a handful of build waves (heap block switches; the knob is read at vgx_create).
`python tests/soak_gpu.py 2000` ran clean on the round-1 build, `... 6000` on the round-2 build (0 mismatches; default,
few-waves and large-sequence contexts in turn)."""
import importlib, sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import pyoracle
from util import assert_mesh_equal
rt = importlib.import_module("vg-renderer_amd.runtime"); wl = importlib.import_module("vg-renderer_amd.workloads")
ctx_default = rt.Context(0)
os.environ["VGX_BUILD_WAVES"] = "3"
ctx_few = rt.Context(0)
os.environ.pop("VGX_BUILD_WAVES", None)
os.environ["VGX_NO_SMALL"] = "1"
ctx_large = rt.Context(0)  # the large-batch launch sequence on these small batches
os.environ.pop("VGX_NO_SMALL", None)
bad = 0
for seed in range(2000, 2000 + int(sys.argv[1] if len(sys.argv) > 1 else 300)):
    ps = wl.fuzz_paths(seed, npaths=64)
    d = wl.fuzz_draws(ps, seed)
    rs = np.random.RandomState(seed)
    d = np.concatenate([d, d[rs.permutation(d.shape[0])]])
    ctx = ctx_few if seed % 3 == 0 else (ctx_large if seed % 3 == 1 else ctx_default)
    ref = pyoracle.tessellate(ps, d)
    pset = rt.PathSet(ctx, ps); dd = rt.upload_draws(d)
    sizes = rt.tessellate_count(ctx, pset, dd, d.shape[0])
    bufs = rt.MeshBuffers(dd.device, sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"])
    rt.tessellate_async(ctx, pset, dd, d.shape[0], bufs); torch.cuda.synchronize()
    class G: pass
    g = G(); g.sizes = sizes
    nv, ni, nm = sizes["num_vertices"], sizes["num_indices"], sizes["num_meshes"]
    g.pos = bufs.pos[:nv].cpu().numpy(); g.color = bufs.color[:nv].cpu().numpy().view(np.uint32)
    g.idx = bufs.idx[:ni].cpu().numpy().view(np.uint16); g.meshes = bufs.meshes[:nm*32].cpu().numpy().view(rt.capi.mesh_dtype)
    try:
        assert int(bufs.dev_status.item()) == 0
        assert_mesh_equal(g, ref, "soak %d" % seed)
    except AssertionError as e:
        bad += 1; print("MISMATCH seed", seed, str(e)[:200])
    pset.close()
print("soak done, mismatches:", bad)
