"""Ablation of the second-generation conv kernel on the block-3 ResConv shape (profiling experiment).
Each configuration runs in a child process because the hook is an environment variable."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import tools.conv_sweep as cs
B = int(sys.argv[1]); v = int(sys.argv[2])
ms = cs.run(B, 272, 480, 64, 64, 1, True, v, reps=6)
print("RESULT", ms)
'''
NAMES = {0: "full", 1: "noDMA", 2: "noEpilogue", 3: "noDMA+noEpi", 7: "MFMA+barriers only (same LDS addr)",
         15: "MFMA only", 4: "fixed LDS addr", 8: "no barrier(!)", 6: "noEpi+fixed LDS addr"}
if __name__ == "__main__":
    B = 8
    flop = 2.0 * B * 272 * 480 * 64 * 64 * 9
    for v in (32, 34):
        for abl in (0, 1, 2, 3, 6, 7, 15):
            env = dict(os.environ, VFI_CONV_ABLATE=str(abl))
            r = subprocess.run([sys.executable, "-c", CHILD % ROOT, str(B), str(v)], env=env, capture_output=True, text=True)
            ms = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            if not ms:
                print(v, abl, "FAILED", r.stderr[-300:])
                continue
            ms = float(ms[0].split()[1])
            print(f"variant {v} ablate {abl:2d} {NAMES.get(abl, ''):36s} {ms:8.4f} ms {flop / ms / 1e9:8.2f} TFLOP/s", flush=True)
