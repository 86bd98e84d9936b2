"""How hipExtStreamCreateWithCUMask's bits map to the chip (8 XCDs x 32 CUs): a grid of resident workgroups on a masked stream reports the XCC id
and HW_ID (shader engine / CU) of the CU each of them landed on (vmmt_probe_where).   python tools/probe_cu_mask.py"""
import collections, ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd import _lib as L
lib = L.lib()
torch.cuda.init()
N = 2048


def where(stream_ptr):
    out = torch.zeros(N, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    L.check(lib.vmmt_probe_where(C.c_void_p(out.data_ptr()), N, 256, 200, C.c_void_p(stream_ptr)), "probe")
    torch.cuda.synchronize()
    v = out.cpu().numpy().astype("uint32")
    xcc = v & 15
    hw = v >> 8
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    per = collections.Counter(int(x) for x in xcc)
    cus = set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    return per, cus


def masked(bits):
    words = [0] * 8
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (C.c_uint32 * 8)(*words)
    out = C.c_void_p()
    L.check(lib.vmmt_stream_create_masked(arr, 8, 0, C.byref(out)), "mask")
    return out.value


plain = torch.cuda.Stream()
for name, bits in (("no mask", None), ("bits 0..191", range(192)), ("bits 0..127", range(128)), ("bits with i % 4 != 3", [i for i in range(256) if i % 4 != 3]),
                   ("bits 0..31", range(32)), ("bits 0..7", range(8)), ("bits 64..255", range(64, 256))):
    st = plain.cuda_stream if bits is None else masked(list(bits))
    per, cus = where(st)
    by_xcc = collections.Counter(c[0] for c in cus)
    print("%-24s distinct CUs %3d   CUs per XCD %s   (se, cu) of XCD 0: %s" % (name, len(cus), [by_xcc.get(x, 0) for x in range(8)],
          sorted((c[1], c[3]) for c in cus if c[0] == 0)[:40]))
