"""A few lines out of a bench.py JSON line (development loop)."""
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("value %.0f samples/s  %.2f us/step (out of idle: %s) | roof %s %.1f us frac %s | board %s W" % (d["value"], d["ms_per_step"] * 1e3, round(d.get("cold_start", {}).get("value", 0)), r["kernel"], r["avg_launch_us"], r["frac"], round((r.get("board_power") or {}).get("watts_mean", 0))))
for k, m in d.get("extras", {}).get("math_modes", {}).items():
    print("  math %s: %.0f samples/s (%.2f of the headline), layer2 %.1f us [%s] frac %s" % (k, m["samples_per_s"], m.get("vs_headline", float("nan")), m["roofline"]["avg_launch_us"], m["roofline"]["bound"], m["roofline"]["frac"]))
if "single_evaluation_in_flight" in d:
    s = d["single_evaluation_in_flight"]; print("single eval %.1f us (8 per replay %s, recorded launches %s), layer2 %.1f us" % (s["us_per_evaluation"], s.get("eight_per_replay", {}).get("us_per_evaluation"), s.get("recorded_launches", {}).get("us_per_evaluation", s.get("recorded_launches")), s["layer2"]["avg_launch_us"]))
e = d.get("extras", {})
for m in e.get("mc_batched_one_minibatch", []):
    print("  S=%d one minibatch: %.0f samples/s %.1f us/eval, layer2 %.1f us frac %s%s" % (m["mc_samples_per_evaluation"], m["samples_per_s"], m["us_per_evaluation"], m["layer2_us_per_launch"], m.get("layer2_frac", m.get("layer2_hbm_frac")), (" | recorded launches %.1f us" % m["us_per_evaluation_recorded_launches"]) if "us_per_evaluation_recorded_launches" in m else ""))
if "lr_variant" in e:
    l = e["lr_variant"]; s1 = l["single_evaluation_in_flight"]
    print("  LR: %.0f samples/s, layer2 %s %.1f us frac %.3f | single %.1f us, layer2 %.1f us frac %.3f" % (l["samples_per_s"], l["roofline"]["kernel"], l["roofline"]["avg_launch_us"], l["roofline"]["frac"], s1["us_per_evaluation"], s1["layer2_us_per_launch"], s1["layer2_hbm_frac"]))
for w in e.get("wide_4096", []):
    r = w["roofline"]
    print("  wide B=%d S=4: %.0f samples/s %.0f us/eval, [%s] %s %.1f us frac %.3f %s" % (w["batch"], w["samples_per_s"], w["us_per_evaluation"], r["bound"], r["kernel"][:40], r["avg_launch_us"], r["frac"], ("sampling %.1f us hbm %.3f" % (r["sampling"]["avg_launch_us"], r["sampling"]["hbm_frac"])) if "sampling" in r else ""))
for c in e.get("c4", []):
    print("  c4 S=%d: %.0f samples/s %.1f us/eval" % (c["mc_samples_per_evaluation"], c["samples_per_s"], c["us_per_evaluation"]))
for t in e.get("training_step", []):
    print("  train step %s S=%d: %.3f ms" % (t["variant"], t["mc_samples"], t["ms_per_step"]))
