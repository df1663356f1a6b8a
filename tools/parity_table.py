"""profiles/r05_parity_benchdims.md from the summaries tests/test_gpu_benchdims.py leaves in gpurun_out/ (run after the GPU tests)."""
import glob, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "profiles", "r05_parity_benchdims.md")
hdr = ["run", "step", "loss (oracle)", "loss err abs", "loss err / (1 + abs ref)", "score err / scale", "loss-grad kernel err (same scores)", "loss-grad shift from the score err (oracle)",
       "grad max err / own max", "/ model max", "grad rms / own max", "ReLU units on other branch", "max abs pre-act of those",
       "NDCG@5 batch-mean delta", "max delta well-cond.", "ill", "top-5 differs (well-cond.)", "full valid order identical"]
lines = ["# Round 5: parity of the benchmarked arithmetic at benchmark dimensions (tests/test_gpu_benchdims.py, MI355X)", "",
         "fp64 oracle evaluated at the engine's weights at every step; gradients on the engine's ReLU branch (units on the other branch counted);",
         "NDCG@5 of the engine's scores vs the oracle's.  `ill` = slates with two of their six best items (different labels) closer than twice the score error.",
         "`loss-grad kernel err` = the loss kernel's d loss / d scores against the oracle's AT THE ENGINE'S OWN SCORES; `shift` = how far the oracle's own",
         "d loss / d scores moves between the engine's scores and the oracle's (the conditioning of the loss; every parameter gradient inherits it).",
         "Both relative to the largest |d loss / d score|.  The loss bar of the tests is RELATIVE, 1e-5 (1 + |ref|): both the absolute error and",
         "that ratio are printed (lambdaLoss sums over the whole batch -- values of 1e3-1e4 --, ListMLE over 1024 items).",
         "Gradient bars are a FIXED table per loss (tests/test_gpu_benchdims.py GRAD_BARS), no longer widened by the measured conditioning shift.", "", "| " + " | ".join(hdr) + " |", "|" + "---|" * len(hdr)]
for f in sorted(glob.glob(os.path.join(R, "gpurun_out", "parity_benchdims_*.json"))):
    name = os.path.basename(f)[len("parity_benchdims_"):-5]
    for r in json.load(open(f)):
        rms = max((v["rms_err"] / v["own_max"] for v in r["grads"].values() if v["own_max"] > 1e-6 * r["grad_model_scale"]), default=0.0)
        e = lambda k: ("%.1e" % r[k]) if k in r else "-"
        lines.append("| %s | %d | %.6g | %.1e | %.1e | %.1e / %.1f | %s | %s | %.1e | %.1e | %.1e | %d / %d | %.1e | %.1e | %.1e | %d | %d (%d) | %d / %d |" % (
            name, r["step"], r["oracle_loss"], r["loss_err"], r["loss_err"] / (1.0 + abs(r["oracle_loss"])), r["score_err"], r["score_scale"], e("lossgrad_kernel_err"), e("lossgrad_shift_from_score_err"),
            r["grad_rel_own_max"], r["grad_rel_model_max"], rms, r["relu_units_on_other_branch"], r["relu_units"], r["max_abs_preact_of_those"],
            r["ndcg5_batch_mean_abs_delta"], r["ndcg5_max_delta_well_conditioned"], r["slates_ill_conditioned"], r["slates_top5_order_differs"],
            r["slates_top5_order_differs_well_conditioned"], r["slates_full_valid_order_identical"], r["slates"]))
open(out, "w").write("\n".join(lines) + "\n")
print("wrote", out, len(lines) - 10, "rows")
