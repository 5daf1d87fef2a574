"""Which event queries fail while a stream capture is in progress?  (torch's ProcessGroupNCCL watchdog thread queries the end
events of eager collectives; the engine captures whole steps holding RCCL kernels.)"""
import threading

import torch

dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
s = torch.cuda.Stream()
res = {}


def q(name, ev):
    def run():
        try:
            res[name] = ev.query()
        except RuntimeError as e:
            res[name] = "ERROR " + str(e).splitlines()[0][:90]
    t = threading.Thread(target=run)
    t.start()
    t.join()
    print(f"{name}: {res[name]}", flush=True)


before = torch.cuda.Event()
with torch.cuda.stream(s):
    x += 1
    before.record()            # recorded eagerly on s, BEFORE s starts capturing
other = torch.cuda.Event()
other.record()                 # recorded eagerly on the default stream
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
inside = torch.cuda.Event()
try:
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        x += 1
        q("eager event of the capturing stream, queried from another thread during the capture", before)
        q("eager event of another stream, queried during the capture", other)
        x += 1
        print("capture still valid after those two queries", flush=True)
        inside.record(torch.cuda.current_stream())
        q("event recorded inside the capture, queried during the capture", inside)
        x += 1
except Exception as e:
    print("capture ended with:", str(e).splitlines()[0][:100])
