import torch, sys
sys.path.insert(0, ".")
from cxrmate_amd import ops
for (B,H,Tq,Tk) in [(32,1,9216,2304),(32,3,2304,576),(32,6,577,145),(32,12,256,256),(32,12,256,576)]:
    D=H*64
    q=torch.randn(B,Tq,D,device="cuda").bfloat16(); k=torch.randn(B,Tk,D,device="cuda").bfloat16(); v=torch.randn(B,Tk,D,device="cuda").bfloat16()
    for _ in range(3): ops.attention(q,k,v,H,0.125,need_lse=True)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.attention(q,k,v,H,0.125,need_lse=True)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*100
    print(B,H,Tq,Tk,"fwd %.1f us %.0f TF/s"%(us, 4.0*B*H*Tq*Tk*64/us/1e6))
