"""Developer probe: host-to-device copy rates of this box -- pageable vs pinned, one and several host threads staging."""
import time
import threading
import numpy as np
import torch

N = 838 * 1024 * 1024 // 4
src = np.random.default_rng(0).random(N, dtype=np.float32)
dst = torch.empty(N, dtype=torch.float32, device="cuda")
tsrc = torch.from_numpy(src)
torch.cuda.synchronize()

def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); a = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - a)
    return best

gb = N * 4 / 1e9
x = t(lambda: dst.copy_(tsrc))
print("pageable -> device, one call           : %6.1f ms  %5.1f GB/s" % (x * 1e3, gb / x))
pin = torch.empty(N, dtype=torch.float32).pin_memory()
x = t(lambda: pin.copy_(tsrc))
print("pageable -> pinned (one host thread)   : %6.1f ms  %5.1f GB/s" % (x * 1e3, gb / x))
x = t(lambda: dst.copy_(pin, non_blocking=True))
print("pinned -> device                       : %6.1f ms  %5.1f GB/s" % (x * 1e3, gb / x))
for nt in (2, 4, 8, 16):
    parts = np.array_split(np.arange(N), nt)
    pn = pin.numpy()
    def work(i):
        a, b = parts[i][0], parts[i][-1] + 1
        np.copyto(pn[a:b], src[a:b])
    def run():
        th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
        [q.start() for q in th]; [q.join() for q in th]
    x = t(run)
    print("pageable -> pinned, %2d host threads     : %6.1f ms  %5.1f GB/s" % (nt, x * 1e3, gb / x))
