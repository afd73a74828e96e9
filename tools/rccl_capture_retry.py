"""Round 4 retry of recording an RCCL collective into a hipGraph (round 3: SIGSEGV inside hipStreamEndCapture,
profiles/r3_rccl_capture_attempt.txt): a pre-warmed communicator (several eager collectives first), each capture error
mode, an all-reduce of the size of config 4's gradient buffer.  Every attempt runs in its own subprocess (a crash must not
take the others down); prints one line per attempt."""
import os
import subprocess
import sys

CHILD = r'''
import os, socket, sys, torch, torch.distributed as dist
mode = sys.argv[1]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
buf = torch.ones(75000, device=dev)
for _ in range(5):
    dist.all_reduce(buf)          # pre-warm: communicator, channels, proxy threads all exist
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    dist.all_reduce(buf)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g, capture_error_mode=mode):
    buf.mul_(2.0)
    dist.all_reduce(buf)
    buf.add_(1.0)
torch.cuda.synchronize()
before = float(buf[0])
g.replay(); g.replay()
torch.cuda.synchronize()
print("CAPTURED mode=%s: value %.1f -> %.1f after two replays (expected x -> 2(2x+1)+1)" % (mode, before, float(buf[0])))
'''


SAGE_CHILD = r'''
import os, socket, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["GGL_REPO"])
os.environ["GGL_SAGE_ONE_GRAPH"] = "1"
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from gammagl_amd import engine
from gammagl_amd.sampler import BlockSampler
from gammagl_amd.synth import rmat_graph
from gammagl_amd.trainer import SAGEBlockTrainer
N, F_in, Hd, C, B = 50000, 32, 64, 7, 512
ei = rmat_graph(N, 600000, seed=2, device=dev)
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn(N, F_in, generator=g, device=dev); y = torch.randint(0, C, (N,), generator=g, device=dev)
bs = BlockSampler(ei, [5, 5], num_nodes=N, eng=engine()); caps = bs.calibrate(B, trials=4, slack=1.5)
tr = SAGEBlockTrainer(bs, F_in, Hd, C, device=dev, caps=caps, world=2, seed=5)
seeds = torch.randperm(N, generator=g, device=dev)[:B].contiguous()
tr.capture(x, y, seeds, warmup=3)
ls = [float(tr.replay()) for _ in range(20)]
torch.cuda.synchronize()
print("CAPTURED sage replica step with its all-reduce as ONE hipGraph: 20 replays, loss %.4f -> %.4f" % (ls[0], ls[-1]))
'''


def main():
    print(f"# torch {__import__('torch').__version__}; one MI355X, world-size-1 NCCL group, pre-warmed communicator")
    for mode in ("global", "thread_local", "relaxed"):
        r = subprocess.run([sys.executable, "-c", CHILD, mode], capture_output=True, text=True, timeout=180,
                           env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        out = [ln for ln in r.stdout.splitlines() if ln.startswith("CAPTURED")]
        if r.returncode == 0 and out:
            print(out[-1])
        else:
            tail = (r.stderr.strip().splitlines() or ["<no stderr>"])[-1][:300]
            print(f"FAILED   mode={mode}: rc {r.returncode} ({'signal ' + str(-r.returncode) if r.returncode < 0 else 'exception'}): {tail}")


A2A_CHILD = CHILD.replace("""    buf.mul_(2.0)
    dist.all_reduce(buf)
    buf.add_(1.0)""", """    buf.mul_(2.0)
    out = torch.empty_like(buf)
    dist.all_to_all_single(out, buf, [buf.numel()], [buf.numel()])
    buf.copy_(out).add_(1.0)""").replace("for _ in range(5):\n    dist.all_reduce(buf)", """for _ in range(5):
    o2 = torch.empty_like(buf); dist.all_to_all_single(o2, buf, [buf.numel()], [buf.numel()])""").replace(
    "with torch.cuda.stream(s):\n    dist.all_reduce(buf)", "with torch.cuda.stream(s):\n    o3 = torch.empty_like(buf); dist.all_to_all_single(o3, buf, [buf.numel()], [buf.numel()])").replace(
    "CAPTURED mode=%s:", "CAPTURED all_to_all_single (split sizes given) mode=%s:").replace("2(2x+1)+1", "2(2x+1)+1, the exchange with itself being the identity")


def a2a():
    for mode in ("thread_local",):
        r = subprocess.run([sys.executable, "-c", A2A_CHILD, mode], capture_output=True, text=True, timeout=180,
                           env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        out = [ln for ln in r.stdout.splitlines() if ln.startswith("CAPTURED")]
        if r.returncode == 0 and out:
            print(out[-1])
        else:
            tail = (r.stderr.strip().splitlines() or ["<no stderr>"])[-1][:300]
            print(f"FAILED   all_to_all_single mode={mode}: rc {r.returncode} ({'signal ' + str(-r.returncode) if r.returncode < 0 else 'exception'}): {tail}")


def sage():
    r = subprocess.run([sys.executable, "-c", SAGE_CHILD], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GGL_REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    out = [ln for ln in r.stdout.splitlines() if ln.startswith("CAPTURED")]
    if r.returncode == 0 and out:
        print(out[-1])
    else:
        tail = (r.stderr.strip().splitlines() or ["<no stderr>"])[-1][:300]
        print(f"FAILED   sage one-graph replica step: rc {r.returncode}: {tail}")


if __name__ == "__main__":
    main()
    a2a()
    sage()
