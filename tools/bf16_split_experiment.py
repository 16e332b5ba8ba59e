"""What a NICE-SLAM decoder would lose if its f32 MFMAs (which run on the VALU's
lanes on gfx950) became bf16 MFMAs on SPLIT operands: weights and
activations are written as sums of bf16 pieces (x = x1 + x2 + ..., each the
bf16 rounding of what is left), the products of pieces are exact in f32, the
MFMA accumulates in f32.  CPU experiment (torch): the colour decoder's shape
(93 Fourier features, 5 x 32 with the fc_c skip terms), 4096 points, largest
error relative to the largest output, against a float64 evaluation.

    python tools/bf16_split_experiment.py   ->  profiles/r06_bf16_split_experiment.txt
"""
import numpy as np, torch
torch.manual_seed(0)
def bf(x): return x.to(torch.bfloat16).to(torch.float32)
def split(x, n):
    parts=[]; r=x.clone()
    for _ in range(n):
        p=bf(r); parts.append(p); r=r-p
    return parts
def mm_split(W, x, nw, nx, maxord):
    Wp=split(W,nw); xp=split(x,nx)
    acc=torch.zeros(W.shape[0], x.shape[1], dtype=torch.float32)
    for i,wi in enumerate(Wp):
        for j,xj in enumerate(xp):
            if i+j<=maxord:
                acc=acc+(wi.double()@xj.double()).float()   # exact products, f32 accumulate per term
    return acc
def mlp(p, c, Ws, mm):
    # NICE-like: embedding sin(B p) 93 -> 32, 5 layers with fc_c skip
    e=torch.sin(Ws['B']@p)
    h=None
    for i in range(5):
        if i==0: a=mm(Ws['W0'],e)+Ws['b'][0]
        elif i==3: a=mm(Ws['W3e'],e)+mm(Ws['Wh'][i],h)+Ws['b'][i]
        else: a=mm(Ws['Wh'][i],h)+Ws['b'][i]
        h=torch.relu(a)+mm(Ws['Wc'][i],c)+Ws['bc'][i]
    return Ws['Wo']@h
N=4096
Ws={'B':torch.randn(93,3)*25,'W0':torch.randn(32,93)*0.2,'W3e':torch.randn(32,93)*0.2,
    'Wh':[None]+[torch.randn(32,32)*0.2 for _ in range(4)],'Wc':[torch.randn(32,32)*0.2 for _ in range(5)],
    'b':torch.randn(5,32,1)*0.2,'bc':torch.randn(5,32,1)*0.2,'Wo':torch.randn(4,32)*0.2}
p=torch.rand(3,N)*2-1; c=torch.randn(32,N)*0.01*30
ref=mlp(p.double(),c.double(),{k:([None if w is None else w.double() for w in v] if isinstance(v,list) else v.double()) for k,v in Ws.items()}, lambda W,x: W@x)
f32=mlp(p,c,Ws,lambda W,x: W@x)
scale=ref.abs().max()
print('f32 vs f64        max rel', float((f32.double()-ref).abs().max()/scale))
for nw,nx,mo in ((2,2,1),(2,2,2),(2,3,2),(3,2,2),(3,3,2),(3,3,3)):
    out=mlp(p,c,Ws,lambda W,x: mm_split(W,x,nw,nx,mo))
    nterms=sum(1 for i in range(nw) for j in range(nx) if i+j<=mo)
    print(f'split W{nw} x{nx} order<={mo} ({nterms} products) vs f64 max rel', float((out.double()-ref).abs().max()/scale), ' vs f32', float((out-f32).abs().max()/scale))
