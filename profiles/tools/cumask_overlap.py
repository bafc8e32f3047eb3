"""can the HBM-bound InstanceNorm passes hide under the (power-limited) MFMA kernels when each gets its own
CU-masked stream (hipExtStreamCreateWithCUMask)?  The overlap study of round 3 used ordinary streams: a persistent conv kernel
holds every CU, so a second stream's kernel only runs in its shadow.  Here: the D-ring conv (8 x 128^3 x 32 -> 32, fp16) on
`NA` CUs and the InstanceNorm + LeakyReLU apply pass of the same tensor on the remaining 256 - NA, alone and together.
Needs the diagnostic build (DGTTA_NCU sizes the persistent grid).  usage: cumask_overlap.py [NA=224] [pattern=block|stride]"""
import ctypes, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
NA = int(sys.argv[1]) if len(sys.argv) > 1 else 224
pattern = sys.argv[2] if len(sys.argv) > 2 else "block"
os.environ.setdefault("DGTTA_LIB", os.path.join(HERE, "libdgtta_hip_diag.so"))
os.environ["DGTTA_NCU"] = str(NA)
import torch
from dg_tta_amd import _lib
from dg_tta_amd._lib import check, ptr
lib = _lib.load()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
DEV = torch.device("cuda:0")
torch.cuda.init()
torch.zeros(1, device=DEV)


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (32 * w + b) in bits) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(s.value, device=DEV)


NB = 256 - NA
if pattern == "block":
    a_bits, b_bits = set(range(NA)), set(range(NA, 256))
else:                    # every (256 / NB)-th CU goes to B
    step = 256 // NB
    b_bits = set(range(step - 1, 256, step))
    a_bits = set(range(256)) - b_bits
sa, sb = masked_stream(a_bits), masked_stream(b_bits)
full = torch.cuda.Stream(DEV)

B, n, c = 8, 128, 32
dt, tdt = 2, torch.float16
x = torch.randn(B, n, n, n, c, device=DEV).to(tdt)
w = torch.randn(c, c, 3, 3, 3, device=DEV) * 0.05
wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(c, c, dt) // 2, dtype=tdt, device=DEV)
check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), c, c, c, c, dt, torch.cuda.current_stream().cuda_stream), "pack")
y = torch.empty((B, n, n, n, c), dtype=tdt, device=DEV)
st = torch.zeros(lib.dgtta_conv3d_stats_bytes(B, c, n, n, n), dtype=torch.uint8, device=DEV)
y2 = torch.randn(B, n, n, n, c, device=DEV).to(tdt)
z2 = torch.empty_like(y2)
mr = torch.empty(B, c, 2, device=DEV)
gamma, beta = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
nws = lib.dgtta_instnorm_ws_bytes(B, c, n ** 3)
ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
torch.cuda.synchronize()


def conv(s):
    check(lib.dgtta_conv3d_k3_fwd(ptr(x), c, ptr(wpack), None, ptr(y), c, ptr(st), B, c, c, c, c, n, n, n, 1, dt, 2, s.cuda_stream), "fwd")


def inorm(s):
    check(lib.dgtta_instnorm_lrelu_fwd(ptr(y2), c, None, ptr(gamma), ptr(beta), ptr(mr), ptr(z2), c, ptr(ws), nws, B, c, n ** 3, 1e-5,
                                       0.01, dt, s.cuda_stream), "in")


def timed(fn, reps):
    for _ in range(40):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


R = 150
os.environ["DGTTA_NCU"] = "256"; lib.dgtta_reload_env()
t_conv_full = timed(lambda: conv(full), R)
t_in_full = timed(lambda: inorm(full), R)
os.environ["DGTTA_NCU"] = str(NA); lib.dgtta_reload_env()
t_conv_a = timed(lambda: conv(sa), R)
t_in_b = timed(lambda: inorm(sb), R // 3)
k = max(1, round(t_conv_a / max(t_in_b, 1e-3) * 1.0))     # IN launches per conv launch that keep both streams busy
def both():
    conv(sa)
    inorm(sb)
t_both = timed(both, R)
print(f"pattern {pattern}, conv on {NA} CUs / InstanceNorm apply on {NB} CUs (8 x 128^3 x 32, fp16)")
print(f"  all 256 CUs, one after the other: conv {t_conv_full:.3f} ms + IN fwd (reduce + apply) {t_in_full:.3f} ms = {t_conv_full + t_in_full:.3f} ms per pair")
print(f"  alone on its masked stream:       conv {t_conv_a:.3f} ms, IN {t_in_b:.3f} ms")
print(f"  both at once (one conv + one IN per round): {t_both:.3f} ms per pair  ->  {(t_conv_full + t_in_full) / t_both:.3f}x the sequential rate")
