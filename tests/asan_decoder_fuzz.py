"""Memory-safety fuzz of vgx_cmdlist_decode (host code, parses bytes from outside): the decoder compiled with g++
-fsanitize=address,undefined and fed valid and malformed streams (nested lists included). Not collected by pytest:
    bash tests/asan_decoder_fuzz.sh
Round 3: 3 120 decodes, no sanitizer report; round 4: + streams with IndexedTriList commands, same."""
import sys, importlib, ctypes as C, numpy as np
sys.path[:0]=['/root/repo','/root/repo/oracle','/root/repo/tests']
import pyvgref as R, frameref as F, cmdlist_util as cu
import test_cmdlist_ref as T
import trilist_frame as TF
rt = importlib.import_module("vg-renderer_amd.runtime")
asan = C.CDLL("/tmp/asan/libcl_asan.so")
asan.vgx_cmdlist_decode.restype = C.c_int
class FakeRt:
    capi = rt.capi
    @staticmethod
    def lib(): return asan
rs = np.random.RandomState(7)
codes = {}
for seed in range(120):
    nchild = seed % 3
    children = [(T.s_random(90000 + 7 * seed + c, top=False), 0) for c in range(nchild)]
    with R.RefContext() as rc:
        lists = {}
        for cs, cf in children:
            h, b = F.record(rc, cs, cf); lists[h] = (b, cf)
        for _ in range(6):
            rc.create_image(8, 8)
        # every third stream: user meshes (IndexedTriList payloads: counts, UV / colour / index arrays) between the paths
        h, data = F.record(rc, TF.s_random(80000 + seed) if (seed % 6 == 0) else T.s_random(80000 + seed, nchildren=nchild))
    # the valid stream first (sanitizer sees the normal paths), then mutations
    r = cu.decode(FakeRt, data, lists=lists); codes[r[0]] = codes.get(r[0], 0) + 1
    for trial in range(25):
        b = bytearray(data)
        k = trial % 4
        if k == 0: b = b[:int(rs.randint(0, len(b) + 1))]
        elif k == 1:
            for _ in range(int(rs.randint(1, 8))): b[int(rs.randint(0, len(b)))] = int(rs.randint(0, 256))
        elif k == 2:
            off = int(rs.randint(0, max(1, len(b) // 4))) * 4
            if off + 4 <= len(b): b[off:off+4] = int(rs.randint(0, 1 << 32, dtype=np.uint64)).to_bytes(4, "little")
        else: b += bytes(rs.randint(0, 256, size=int(rs.randint(1, 64))).astype(np.uint8))
        r = cu.decode(FakeRt, bytes(b), lists=lists); codes[r[0]] = codes.get(r[0], 0) + 1
print("asan/ubsan fuzz done:", codes)
