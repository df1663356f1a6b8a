"""ONE staged run (like profiles/r04_reference_on_gpu_box.md): the UNMODIFIED allrank/main.py on the GPU box under the real launcher --
`allrank_amd.launch.main([...])`, two ranks (gloo, both on the box's one GPU), real `fit()` / FusedTrainer -- against the same job as one
rank at the same global batch.  Needs ALLRANK_REFERENCE (a checkout staged in an ignored scratch directory for the one call) and a GPU.

    ALLRANK_REFERENCE=$GRAFT_REPO_ROOT/.ref_stage python tools/staged_launcher_check.py  -> gpurun_out/staged_launcher_run.md

(The reference's optional dependencies that this image lacks -- torchvision, gcsfs, tensorboardX, flatten_dict -- are stubbed by
oracle/ref_loader.py, test infrastructure; that is why a rank is started as `python -c "load_reference(); launch.main(argv)"` instead of
`python -m allrank_amd.launch`: same function, stubs installed first.)"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

RANK_CMD = ("import sys; sys.path.insert(0, %r); from oracle.ref_loader import load_reference; load_reference(stable_sort=False); "
            "import logging; logging.basicConfig(level=logging.INFO); from allrank_amd import launch; sys.exit(launch.main(sys.argv[1:]))" % ROOT)


def main():
    from pathlib import Path
    from allrank_amd import launch
    from tests.test_reference_main import CONFIG, _prepare
    tmp = Path(tempfile.mkdtemp(prefix="staged_launcher_"))
    argv = _prepare(tmp)
    job = argv[1:3]
    cfgs = {}
    for bs, loss in ((16, "listNet"), (32, "listNet"), (16, "neuralNDCG"), (32, "neuralNDCG")):
        cfg = json.loads(json.dumps(CONFIG))
        cfg["data"]["path"] = str(tmp / "dummy_data")
        cfg["data"]["batch_size"] = bs
        cfg["loss"] = {"name": loss, "args": {}}
        path = tmp / ("cfg_%s_b%d.json" % (loss, bs))
        path.write_text(json.dumps(cfg))
        cfgs[(loss, bs)] = str(path)
    out = ["# Staged run: the unmodified `allrank/main.py` under `allrank_amd.launch` on the GPU box (round 5)", "",
           "`tools/staged_launcher_check.py` with `ALLRANK_REFERENCE` pointing at a checkout staged in the ignored scratch directory `.ref_stage/` for this",
           "one call (never committed, deleted afterwards; nothing the driver runs reads it).  Each job = `allrank.main.run()` exactly as the reference",
           "ships it (`main.py:34-110`): libsvm loading, `create_data_loaders`, `make_model`, `CustomDataParallel` branch, optimizer / scheduler, `fit`,",
           "`dump_experiment_result`, `assert_expected_metrics` -- started by `allrank_amd.launch.main([...])`.  Two ranks share this box's one MI355X",
           "(gloo; RCCL needs one GPU per rank), `batch_size` 16 per rank; the one-rank job uses `batch_size` 32: the same global batches.", "",
           "device: %s, torch %s" % (torch.cuda.get_device_name(0), torch.__version__), ""]
    ok = True
    for loss in ("listNet", "neuralNDCG"):
        res = {}
        for world, bs in ((1, 32), (2, 16)):
            run_id = "%s_w%d" % (loss, world)
            logs = tmp / ("logs_" + run_id)
            rc = launch.spawn(world, [sys.executable, "-c", RANK_CMD, "--"] + job + ["--run-id", run_id, "--config-file-name", cfgs[(loss, bs)]],
                              backend="gloo" if world > 1 else None, devices=[0] * world if world > 1 else None, timeout=900, log_dir=str(logs))
            text = "".join(open(logs / f).read() for f in sorted(os.listdir(logs)))
            rdir = os.path.join(job[1], "results", run_id)
            res[world] = dict(rc=rc, fused="allrank_amd.fit: fused step" in text, total_batch=[l.split("total batch size is")[1].strip()[:30] for l in text.splitlines() if "total batch size is" in l][:2],
                              result=json.load(open(os.path.join(rdir, "experiment_result.json"))) if rc == 0 else None,
                              weights=torch.load(os.path.join(rdir, "model.pkl"), map_location="cpu") if rc == 0 else None,
                              epochs=[l.split("Epoch :")[1].strip()[:160] for l in text.splitlines() if "Epoch :" in l and "allrank_amd.fit" in l][:6], tail=text[-1500:])
        a, b = res[1], res[2]
        out += ["## loss `%s`" % loss, ""]
        for world in (1, 2):
            r = res[world]
            out += ["* %d rank(s): exit code %d, fused step: %s, loader log: %s" % (world, r["rc"], r["fused"], r["total_batch"])]
            for e in r["epochs"][:3]:                      # (rank 0's log comes first)
                out += ["    - Epoch : " + e]
        if a["rc"] or b["rc"]:
            ok = False
            out += ["", "FAILED", "```", a["tail"], b["tail"], "```"]
            continue
        ra, rb = a["result"], b["result"]
        keys = [k for k in ra if k.startswith(("train_metrics", "val_metrics"))]
        werr = max(float((a["weights"][k] - b["weights"][k]).abs().max()) for k in a["weights"])
        n_ok = sum(int(((a["weights"][k] - b["weights"][k]).abs() <= 5e-5).sum()) for k in a["weights"])
        n_all = sum(v.numel() for v in a["weights"].values())
        out += ["", "| | 1 rank (batch 32) | 2 ranks (2 x 16) |", "|---|---|---|"]
        for k in keys + ["num_params", "epochs"]:
            out += ["| %s | %s | %s |" % (k, ra[k], rb[k])]
        out += ["", "trained weights (`model.pkl`, same `state_dict` keys: %s): max abs difference %.2e, %.1f %% of the entries within 5e-5 (12 Adam steps of lr 1e-3 / 5e-4;"
                % (set(a["weights"]) == set(b["weights"]), werr, 100.0 * n_ok / n_all),
                "entries whose gradient is below its round-off move by lr x sign(noise) in any arithmetic)", ""]
        good = (a["fused"] and b["fused"] and all(abs(ra[k] - rb[k]) <= 2e-3 for k in keys) and n_ok >= 0.9 * n_all
                and not any(k.startswith("module.") for k in b["weights"]))
        out += ["**%s**" % ("agree" if good else "DISAGREE"), ""]
        ok = ok and good
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "staged_launcher_run.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
