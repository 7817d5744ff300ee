#!/usr/bin/env python
"""Is the GEMM wall clock set by the kernel's cycles or by the chip's power management?  Runs the c_fc GEMM (cfg 8) in a
sustained loop on (a) N(0,1) activations x N(0,1/K) weights - the bench's data -, (b) zero-filled operands, (c) small-integer
operands, while a sampler thread reads the shader clock and the socket power from the amdgpu hwmon / rocm-smi; prints
ms per launch and the sampled sclk / power for each fill.  (MI355X_MICROARCH.md 'DVFS give-back'.)"""
import glob
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch  # noqa: E402
from vitlens_hip import ops  # noqa: E402


def read_first(paths):
    for p in paths:
        try:
            return open(p).read().strip()
        except OSError:
            pass
    return None


def sample():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        f = read_first([hw + "/freq1_input"]); pw = read_first([hw + "/power1_input", hw + "/power1_average"])
        if f:
            out["sclk_mhz"] = int(f) / 1e6
        if pw:
            out["power_w"] = int(pw) / 1e6
        if out:
            return out
    for card in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        for l in open(card).read().splitlines():
            if l.strip().endswith("*"):
                out["sclk_level"] = l.strip()
                return out
    return out


def smi():
    try:
        return subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:          # noqa: BLE001
        return f"rocm-smi failed: {e}"


def main():
    M, N, K = 65536, 4096, 1024
    cfg = int(os.environ.get("KB_CFG", "8"))
    fills = {"randn": lambda *s: torch.randn(*s, device="cuda"), "zeros": lambda *s: torch.zeros(*s, device="cuda"),
             "smallint": lambda *s: torch.randint(-2, 3, s, device="cuda").float()}
    print("idle:", sample())
    for name, mk in fills.items():
        a = mk(M, K).bfloat16(); w = (mk(N, K) * (K ** -0.5 if name == "randn" else 1.0)).bfloat16()
        bias = torch.zeros(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for label, kw in (("bf16", {}), ("gelu", {"act": ops.ACT_GELU})):
            f = lambda: ops.gemm(a, w, bias, out=out, epi=ops.EPI_BF16, cfg=cfg, **kw)
            for _ in range(20):
                f()
            torch.cuda.synchronize()
            samples, stop = [], False

            def run():
                while not stop:
                    samples.append(sample()); time.sleep(0.05)
            th = threading.Thread(target=run); th.start()
            n = 1500
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                f()
            e1.record(); torch.cuda.synchronize()
            stop = True; th.join()
            ms = e0.elapsed_time(e1) / n
            sk = [s["sclk_mhz"] for s in samples if "sclk_mhz" in s]; pw = [s["power_w"] for s in samples if "power_w" in s]
            print(f"{name:9s} {label:5s} cfg{cfg}: {ms:.4f} ms/launch {2.0 * M * N * K / ms / 1e9:7.1f} TF/s | sclk MHz "
                  f"{(sum(sk) / len(sk)) if sk else float('nan'):.0f} (min {min(sk) if sk else 0:.0f}) power W {(sum(pw) / len(pw)) if pw else float('nan'):.0f} "
                  f"(max {max(pw) if pw else 0:.0f}) samples {len(samples)} {samples[-1] if samples and not sk else ''}", flush=True)
        if name == "randn":
            print(smi()[-1500:])
        del a, w, out


if __name__ == "__main__":
    main()
