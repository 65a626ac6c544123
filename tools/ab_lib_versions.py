"""A/B builds of libcmfhip.so from different commits in ONE process on one device, launches interleaved.

    python tools/ab_lib_versions.py [m,d,p,k] [rounds] [--pmc]

Every ``tools/ab/libcmfhip_<tag>.so`` (built out of a git worktree of that commit with the command of pycmf_amd/build.py; see
tools/README.md) plus the working tree's ``pycmf_amd/libcmfhip.so`` (tag HEAD) gets its own context holding the same
synthetic C4 problem; each round runs, for every library in turn, the V-partial launches of one MU iteration
(``cmf_mu_v_partials``: the TN pass X^T U, the NN pass Y Z, the Gram) and ``cmf_mu_uz_update`` (NN pass X V, TN pass Y^T V)
with HIP events around every launch.  Prints per library the median / min launch time of the TN and NN data passes.

Under ``rocprofv3 --pmc FETCH_SIZE --kernel-trace`` the dispatch order is the launch order printed with --pmc: library by
library inside each round, so the counter rows can be attributed to a library by position (tools/ab_lib_pmc.py).
"""
import ctypes as C
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
m, d, p, k = (int(x) for x in (args[0] if args else "65536,65536,65536,256").split(","))
rounds = int(args[1]) if len(args) > 1 else 10
pmc = "--pmc" in sys.argv

libs = [("HEAD", os.path.join(ROOT, "pycmf_amd", "libcmfhip.so"))]
for path in sorted(glob.glob(os.path.join(ROOT, "tools", "ab", "libcmfhip_*.so"))):
    libs.append((os.path.basename(path)[len("libcmfhip_"):-3], path))

vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double


class Lib:
    def __init__(self, tag, path):
        self.tag = tag
        self.l = C.CDLL(path, mode=os.RTLD_LOCAL)
        self.l.cmf_last_error.restype = C.c_char_p
        self.ctx = vp()
        self.chk(self.l.cmf_ctx_create(C.byref(self.ctx), i32(0), vp(None)))
        self.chk(self.l.cmf_set_problem(self.ctx, i64(m), i64(d), i64(p), i32(k)))
        self.chk(self.l.cmf_fill_data_synthetic(self.ctx, i32(0), C.c_uint64(42), i64(0), i64(0)))
        self.chk(self.l.cmf_fill_data_synthetic(self.ctx, i32(1), C.c_uint64(43), i64(0), i64(0)))
        sc = (0.7979 / k) ** 0.5
        for w in range(3):
            self.chk(self.l.cmf_fill_factor_synthetic(self.ctx, i32(w), C.c_uint64(101 + w), i64(0), dbl(sc)))
        n = i64()
        self.chk(self.l.cmf_v_buf_elems(self.ctx, C.byref(n)))
        self.buf = vp()
        self.chk(self.l.cmf_scratch_alloc(self.ctx, i64(n.value * 4), C.byref(self.buf)))
        self.tn, self.nn = [], []

    def chk(self, rc):
        if rc:
            raise RuntimeError("%s: %s" % (self.tag, self.l.cmf_last_error().decode()))

    def passes(self):
        self.chk(self.l.cmf_mu_v_partials(self.ctx, self.buf))
        self.chk(self.l.cmf_mu_uz_update(self.ctx, dbl(0.0), dbl(0.0), i32(5)))

    def timed_round(self):
        self.chk(self.l.cmf_kernel_timing(self.ctx, i32(1)))
        self.chk(self.l.cmf_kernel_timing_reset(self.ctx))
        self.passes()
        for cls, dst in ((1, self.tn), (0, self.nn)):
            ms, n, fl = dbl(), i64(), dbl()
            self.chk(self.l.cmf_kernel_time(self.ctx, i32(cls), C.byref(ms), C.byref(n), C.byref(fl)))
            dst.append((ms.value / max(n.value, 1), fl.value / max(ms.value, 1e-9) / 1e9))
        self.chk(self.l.cmf_kernel_timing(self.ctx, i32(0)))


objs = [Lib(t, pth) for t, pth in libs]
for o in objs:            # warm-up: workspaces, code objects
    o.passes()
    o.chk(o.l.cmf_sync(o.ctx))
print("launch order per round:", " ".join(o.tag for o in objs), "(each: TN X^T U, NN Y Z, gram, gram, NN X V, update, TN Y^T V, update)")
for r in range(rounds):
    order = objs if r % 2 == 0 else objs[::-1]   # alternate the order: no library always runs behind the same predecessor
    if pmc:
        order = objs
    for o in order:
        if pmc:
            o.passes()
            o.chk(o.l.cmf_sync(o.ctx))
        else:
            o.timed_round()
if not pmc:
    med = lambda a: sorted(a)[len(a) // 2]
    for o in objs:
        tn_ms, tn_tf = [x[0] for x in o.tn], [x[1] for x in o.tn]
        nn_ms, nn_tf = [x[0] for x in o.nn], [x[1] for x in o.nn]
        print("%-10s TN %.3f ms median (min %.3f, max %.3f) %.1f TF/s | NN %.3f ms median (min %.3f, max %.3f) %.1f TF/s  [%d rounds x 2 launches]"
              % (o.tag, med(tn_ms), min(tn_ms), max(tn_ms), med(tn_tf), med(nn_ms), min(nn_ms), max(nn_ms), med(nn_tf), len(tn_ms)))
for o in objs:
    o.l.cmf_ctx_destroy(o.ctx)
