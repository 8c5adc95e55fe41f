#!/usr/bin/env python
"""Print the DESIGN.md section-5 table of a round from profiles/<tag>_bench.json (+ PMC files): python tools/exp/r05_design_table.py r05"""
import json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles")
d = json.load(open(os.path.join(P, f"{tag}_bench.json")))
def mfma(cfg):
    f = os.path.join(P, f"{tag}_pmc_mfma.json" if cfg is None else f"{tag}_pmc_mfma_{cfg}.json")
    if not os.path.exists(f):
        return {}
    return {k: v["mfma_busy"] for k, v in json.load(open(f))["families"].items()}
def row(name, value, unit, step, fam, whole, mb):
    g, a, n = fam["gemm"], fam["attention"], fam["groupnorm"]
    gm = f", {mb['gemm']:.3f}" if "gemm" in mb else ""
    am = f", {mb['attention']:.3f}" if "attention" in mb else ""
    st = f"{step:.2f} ms" if step else "—"
    print(f"| {name} | **{value:.2f} {unit}** | {st} | {g['tflops']:.0f} ({g['frac']:.3f}){gm} | {a['tflops']:.0f} ({a['frac']:.3f}){am} | {n['gbs']:.0f} | {whole['achieved']:.0f} TF/s ({whole['frac']:.3f}) |")
print("| Config | value | per UNet step | GEMM family TF/s (frac of 2.5 PF), MFMA busy | attention TF/s (frac), MFMA busy | GroupNorm GB/s | whole path |")
print("|---|---|---|---|---|---|---|")
row("**SDv2 512², DDIM-50, batch 1 (headline)**", d["value"], "latents/s", d["per_unet_step_ms"], d["roofline"]["families"], d["roofline"]["whole_path"], mfma(None))
names = {"wukong_512_plms": "Wukong 512², PLMS-50, batch 8", "sd2_768": "SDv2 768², DDIM-50, 4 per GPU", "glide_256": "GLIDE 256², 60 + 27 steps, 8 per GPU"}
for k, v in d["other_configs"].items():
    row(names[k], v["value"], v.get("unit", "latents/s" if k != "glide_256" else "images/s"), v.get("per_unet_step_ms"), v["families"], v["whole_path"], mfma(k) if k != "glide_256" else {})
cb = d.get("cpu_baseline")
if cb:
    print(f"| CPU oracle ({cb['cores']} threads), SDv2 B = 1 eval | {cb['sample'].split(':')[1].split('(')[0].strip()} ⇒ {cb['value']:.4f} latents/s | | | | | |")
t = d["roofline"].get("traffic")
print("traffic per GEMM-family launch:", t, d["roofline"].get("traffic_note", ""))
f = os.path.join(P, f"{tag}_pmc_traffic.json")
if os.path.exists(f):
    j = json.load(open(f)); print("PMC per evaluation: read MB", j["total_read_MB_per_eval"], "write MB", j["total_write_MB_per_eval"], {k: v["launches_per_eval"] for k, v in j["families"].items()})
