"""dev (round 5): does the teacher forward get faster as TWO half-batch pipelines on two streams, each persistent kernel on half of the
CUs (DKD_CU_LIMIT=128), than as one full-batch pipeline?  Tile boundaries of the two halves are not in step, so one half's store
epilogues run under the other's K loops (profiles/r05_wide_kernel_half_of_the_workgroups_skip_their_stores.txt says a third of the
epilogue is the chip-wide write burst).
usage: [DKD_CU_LIMIT=128] python tools_dev/teacher_halves_probe.py full|halves"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deltakd_amd import vit

mode = sys.argv[1]
dev = "cuda:0"
torch.manual_seed(42)
t1 = vit.create_model("deit_base_distilled_patch16_224", num_classes=1000).to(dev).eval()
for p in t1.parameters():
    p.requires_grad = False
t2 = copy.deepcopy(t1) if mode == "halves" else None
x = torch.randn(256, 3, 224, 224, device=dev)
xa, xb = x[:128].contiguous(), x[128:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
taps = (0, 1, 11)


def step():
    with torch.no_grad():
        if mode == "full":
            t1.forward_with_taps(x, taps, head=False)
        else:
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                t1.forward_with_taps(xa, taps, head=False)
            with torch.cuda.stream(s2):
                t2.forward_with_taps(xb, taps, head=False)
            cur.wait_stream(s1)
            cur.wait_stream(s2)


for _ in range(4):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 12
for _ in range(n):
    step()
e1.record()
torch.cuda.synchronize()
print(f"{mode} (DKD_CU_LIMIT={os.environ.get('DKD_CU_LIMIT', '-')}): {e0.elapsed_time(e1) / n:.3f} ms per 256 images")
