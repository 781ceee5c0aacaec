"""Dev tool (diagnostic build with -DT2_STAMPS only): where one LSTM step launch spends its time."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _devlib import SMA, LSA, decoder_setup
from tacotron2_subword_amd import _lib as L, ops
hp, P, dims, W = decoder_setup(SMA)
B, T = 64, 24
L.set_precision(os.environ.get("T2_PREC", "bf16"))
mem = torch.randn(B, 100, 512, device="cuda") * .5; mems = torch.randn(B, 60, 512, device="cuda") * .5
mels = torch.randn(B, 80, T, device="cuda"); tl = torch.full((B,), 100, device="cuda"); bl = torch.full((B,), 60, device="cuda")
for _ in range(3):
    dp = ops.decoder_forward(W, dims, mem, mems, tl, bl, mels, training=True, prenet_dropout=True, seed=1)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
L.lib().t2_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
print("rc", L.lib().t2_debug_read_stamps(buf, 16))
for name, o in (("att (2 streams)", 0), ("dec", 8)):
    v = [buf[o + i] for i in range(5)]
    print(name, "entry->staged %.2f us | K loop %.2f us | epilogue %.2f us | query partials %.2f us | total %.2f us" % tuple(
        [(v[i + 1] - v[i]) / 100.0 for i in range(4)] + [(v[4] - v[0]) / 100.0]))
    print("   probes: first weight elem %.2f us | first activation elem %.2f us | second of each %.2f us | rest of fill %.2f us" % (
        (buf[o + 5] - v[0]) / 100.0, (buf[o + 6] - buf[o + 5]) / 100.0, (buf[o + 7] - buf[o + 6]) / 100.0, (v[1] - buf[o + 7]) / 100.0))

# attention kernels (workgroup b=7, phone stream, split 0); needs a backward pass too
dmel = torch.randn(B, T, 80, device="cuda"); dgate = torch.randn(B, T, device="cuda")
G, dm, dms = ops.decoder_backward(W, P, dims, dp, mem, mems, dmel, dgate, training=True, prenet_dropout=True, seed=1)
torch.cuda.synchronize()
buf2 = (C.c_ulonglong * 32)()
L.lib().t2_debug_read_stamps_attn.argtypes = [C.c_void_p, C.c_int]
print("rc", L.lib().t2_debug_read_stamps_attn(buf2, 32))
v = [buf2[i] for i in range(6)]
print("attention fwd: query %.2f | energies %.2f | recurrence %.2f | context loads %.2f | ctx reduce+stores %.2f | total %.2f us" % tuple(
    [(v[i + 1] - v[i]) / 100.0 for i in range(5)] + [(v[5] - v[0]) / 100.0]))
v = [buf2[8 + i] for i in range(6)]
print("attention bwd: dctx assembly %.2f | g pass %.2f | de/carry %.2f | energies bwd %.2f | reduce+stores %.2f | total %.2f us" % tuple(
    [(v[i + 1] - v[i]) / 100.0 for i in range(5)] + [(v[5] - v[0]) / 100.0]))
