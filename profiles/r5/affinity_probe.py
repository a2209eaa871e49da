import os, sys, subprocess
print("before torch", len(os.sched_getaffinity(0)))
import torch
print("after torch", len(os.sched_getaffinity(0)))
x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
print("after cuda init", len(os.sched_getaffinity(0)))
print(subprocess.run(["bash", "-c", "taskset -p $$; nproc"], capture_output=True, text=True).stdout)
