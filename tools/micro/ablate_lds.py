import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build(); ge.load_package()
from cfi_amd import _lib
lib = _lib.load(); _lib.check(lib.vfi_init(0), "init")
def run(n,h,w,c,variant,slope,reps=6):
    x = torch.rand(n,h,w,c,device="cuda")-0.5
    wt = (torch.rand(c,c,3,3)-0.5)*0.1; b = torch.rand(c)-0.5
    out = torch.empty(n,h,w,c,device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    lib.vfi_conv3x3(p(x),p(wt),p(b),None,p(out),n,h,w,c,c,1,1,slope,variant,None)
    lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
    for _ in range(reps): lib.vfi_conv3x3(p(x),p(wt),p(b),None,p(out),n,h,w,c,c,1,1,slope,variant,None)
    lib.vfi_trace_enable(0)
    calls, tot = list(_lib.trace_report().values())[0]
    ms = tot/calls
    print(f"variant {variant} c={c} slope={slope}: {ms*1e3:.1f} us  {2*n*h*w*c*c*9/ms/1e9:.1f} TFLOP/s", flush=True)
for slope in (0.2, 0.123):
    run(8,272,480,64,32,slope)
    run(8,136,240,96,35,slope)
    run(8,68,120,128,34,slope)
