import torch, time, json
dev="cuda:0"
shapes={"dWx1":(2048,448,32768),"dWh1":(2048,512,32768),"dWx2":(1024,512,32768),"dWh2":(1024,256,32768),"dWfc":(256,704,32768),
        "xproj1":(32768,2048,448),"dense_fwd":(32768,704,256),"dense_bwd":(32768,256,704)}
out={}
for name,(M,N,K) in shapes.items():
    A=torch.randn(M,K,device=dev,dtype=torch.bfloat16); B=torch.randn(N,K,device=dev,dtype=torch.bfloat16)
    for _ in range(3): C=A@B.t()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): C=A@B.t()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/20*1e3
    out[name]={"us":round(us,1),"TF":round(2*M*N*K/us/1e6,1)}
print(json.dumps(out))
