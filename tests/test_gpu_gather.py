"""vgx_gather_sizes / vgx_gather (the RCCL gather behind the C-ABI) driven from C++ (tests/native/gather_test.cpp):
  - 1 rank over the REAL librccl (binding by dlopen, ncclAllGather, local copies, capacity check),
  - 2 and 3 ranks as threads on the one GPU over tests/native/fake_rccl.cpp (RCCL refuses two ranks on one device): shard ->
    tessellate -> gather -> byte-identical streams and mesh table to a single context tessellating the whole batch,
    root = first and last rank (so the root's own block is rebased too).
The real multi-GPU transfer is only exercised by the driver's N-GPU bench (bench.py --gpus N)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vg-renderer_amd")


@pytest.fixture(scope="module")
def binaries(tmp_path_factory):
    d = tmp_path_factory.mktemp("gather")
    exe, fake = str(d / "gather_test"), str(d / "libfake_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "gather_test.cpp"),
                           "-L", PKG, "-lvgx", "-L/opt/rocm/lib", "-lrccl", "-ldl", "-lpthread", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    subprocess.check_call(["g++", "-shared", "-fPIC", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "native", "fake_rccl.cpp"),
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", fake])
    return exe, fake


def test_gather_one_rank_real_rccl(binaries):
    exe, _ = binaries
    out = subprocess.check_output([exe, "real"], text=True, timeout=300)
    assert "ranks 1 root 0" in out and "identical to the single-context run" in out


@pytest.mark.parametrize("nranks,ndraws", [(2, 1000), (3, 1000), (4, 3), (4, 7)])
def test_gather_threaded_ranks(binaries, nranks, ndraws):
    """(4, 3): fewer draws than ranks -- empty shards send and receive nothing and still take part in the size exchange."""
    exe, fake = binaries
    out = subprocess.check_output([exe, "fake", fake, str(nranks), str(ndraws)], text=True, timeout=300)
    assert out.count("identical to the single-context run") == 2, out
    assert "MISMATCH" not in out


@pytest.mark.parametrize("nranks,ndraws,tiles", [(2, 4000, 3), (3, 2000, 2), (4, 5, 2)])
def test_gather_at_tiled_frames(binaries, nranks, ndraws, tiles):
    """vgx_gather_at: every rank cuts its draws into tiles, tessellates them one after the other and gathers tile by tile
    (what lets the gather of tile t overlap the tessellation of tile t + 1); with VGX_GATHER_CHUNK_MB=1 every transfer
    leaves in several pieces. The gathered frame must be byte-identical to a single context tessellating everything."""
    exe, fake = binaries
    env = dict(os.environ, VGX_GATHER_CHUNK_MB="1")
    out = subprocess.check_output([exe, "tiles", fake, str(nranks), str(ndraws), str(tiles)], text=True, timeout=300, env=env)
    assert out.count("identical to the single-context run") == 2, out
    assert "MISMATCH" not in out


def test_gather_eight_ranks_into_a_70_gb_destination(binaries):
    """The root of BASELINE config 4 (Tiger x80k over 8 GPUs) receives 8 x 8.75 GB. Eight thread-ranks on the one GPU send the
    same 8.75 GB local block (one tessellation, shared) into one 70 GB destination: 64-bit offsets, capacity checks and the
    piece splitting (default 256 MiB) at full size; first / last megabyte of every block and the rebased mesh records checked."""
    import torch
    free, total = torch.cuda.mem_get_info(0)
    if free < 110 * (1 << 30):
        pytest.skip("needs ~95 GB of free device memory")
    exe, fake = binaries
    out = subprocess.check_output([exe, "big", fake, "8", "8.75"], text=True, timeout=900)
    assert "blocks identical, mesh records rebased" in out and "MISMATCH" not in out, out
