import sys, shutil, subprocess
for lib in sys.argv[1:]:
    shutil.copy(lib, "sqfa_amd/lib/libsqfa_hip.so")
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.'); import torch, time_pairs as t; t.run(1000,16,False,torch.float64,reps=4); t.run(600,16,True,torch.float64,reps=4); t.run(300,32,False,torch.float64,reps=3)"], capture_output=True, text=True)
    print(lib); print("\n".join(l for l in out.stdout.splitlines() if l.startswith("C=")) or out.stderr[-500:], flush=True)
