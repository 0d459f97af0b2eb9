import torch, time
shapes=[(63040,2304,768),(63040,768,768),(63040,3072,768),(63040,768,3072),(8192,8192,8192)]
for M,N,K in shapes:
    A=torch.randn(M,K,device="cuda",dtype=torch.bfloat16); W=torch.randn(N,K,device="cuda",dtype=torch.bfloat16)
    for _ in range(3): C=A@W.t()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): C=A@W.t()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(M,N,K,"%.1f us %.0f TF"%(ms*1e3, 2*M*N*K/ms/1e9))
