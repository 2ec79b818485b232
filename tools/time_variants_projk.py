import sys, shutil, subprocess
for lib in sys.argv[1:]:
    shutil.copy(lib, "sqfa_amd/lib/libsqfa_hip.so")
    out = subprocess.run([sys.executable, "tools/time_projection_kernel.py"], capture_output=True, text=True)
    print(lib); print("\n".join(l for l in out.stdout.splitlines() if l.startswith("C=")), flush=True)
