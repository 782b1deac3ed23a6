import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.multiprocessing as mp
import test_dp_gpu as T

def w(rank, world, port):
    try:
        T._worker(rank, world, port, {})
        print("rank", rank, "ok", flush=True)
    except Exception:
        print("RANK", rank, "FAILED", flush=True); traceback.print_exc(); sys.stderr.flush()

if __name__ == "__main__":
    mp.spawn(w, args=(2, T._free_port()), nprocs=2, join=True)
