import torch, sys, math
sys.path.insert(0,'/root/repo')
from oracle import vit_ref
import torch.nn.functional as F
torch.manual_seed(3)
m = vit_ref.create_model_ref("deit_base_distilled_patch16_224", 1000, 0.0).eval()
with torch.no_grad():
    for blk in m.blocks: blk.mlp.fc2.weight.mul_(8.0)
bf = lambda t: t.to(torch.bfloat16).float()
x_img = torch.randn(2,3,224,224)
def run(mode):
    # mode: 'fp32', 'std' (bf16 LN output, bf16 W), 'fold' (bf16 x, bf16(gamma*W), stats from f32 x one-pass)
    with torch.no_grad():
        x = m.patch_embed(x_img) if hasattr(m,'patch_embed') else None
        B = x.shape[0]
        toks = [m.cls_token.expand(B,-1,-1)]
        if hasattr(m,'dist_token'): toks.append(m.dist_token.expand(B,-1,-1))
        x = torch.cat(toks+[x],1) + m.pos_embed
        taps=[]
        for blk in m.blocks:
            def ln_lin(x, norm, lin):
                W, b = lin.weight, lin.bias
                if mode=='fp32':
                    return F.linear(F.layer_norm(x,(x.shape[-1],),norm.weight,norm.bias,1e-6), W, b)
                if mode=='std':
                    y = bf(F.layer_norm(x,(x.shape[-1],),norm.weight,norm.bias,1e-6))
                    return y @ bf(W).t() + b
                D = x.shape[-1]
                s1 = x.sum(-1,keepdim=True); s2=(x*x).sum(-1,keepdim=True)
                mu = s1/D; var = s2/D - mu*mu; rstd = torch.rsqrt(var+1e-6)
                Wp = bf(norm.weight[None,:]*W)
                c = Wp.sum(1)
                bp = W @ norm.bias + b
                acc = bf(x) @ Wp.t()
                return rstd*(acc - mu*c) + bp
            a = blk.attn
            qkv = ln_lin(x, blk.norm1, a.qkv)
            if mode!='fp32': qkv = bf(qkv)
            Bq,N,_ = qkv.shape
            H = a.num_heads
            qkv = qkv.reshape(Bq,N,3,H,-1).permute(2,0,3,1,4)
            o = F.scaled_dot_product_attention(qkv[0],qkv[1],qkv[2]).transpose(1,2).reshape(Bq,N,-1)
            if mode!='fp32': o = bf(o); x = x + (o @ bf(a.proj.weight).t() + a.proj.bias)
            else: x = x + a.proj(o)
            h = ln_lin(x, blk.norm2, blk.mlp.fc1)
            h = F.gelu(h)
            if mode!='fp32': h = bf(h); f = h @ bf(blk.mlp.fc2.weight).t() + blk.mlp.fc2.bias
            else: f = blk.mlp.fc2(h)
            taps.append(f)
            x = x + f
        return taps, x
t32,x32 = run('fp32'); ts,xs = run('std'); tf_,xf = run('fold')
rel = lambda a,b: ((a-b).norm()/b.norm()).item()
for i in (0,1,5,11):
    print(i, 'std', rel(ts[i],t32[i]), 'fold', rel(tf_[i],t32[i]), 'fold-vs-std', rel(tf_[i], ts[i]))
print('x final std', rel(xs,x32), 'fold', rel(xf,x32))
xm = x32.mean(-1).abs().mean().item(); xsd = x32.std(-1).mean().item(); print('mean/std of residual rows', xm, xsd)
