"""N GPUs of one node under an UNMODIFIED allrank/main.py: one process per GPU instead of the reference's nn.DataParallel.

    python -m allrank_amd.launch --nproc 8 -- --job-dir JOB --run-id RUN --config-file-name CONFIG.json

starts 8 ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment; under ``torchrun``
the same module is the per-rank entry point: ``torchrun --nproc-per-node 8 -m allrank_amd.launch -- <main.py args>``).  Each rank

  1. binds its GPU (``torch.cuda.set_device(LOCAL_RANK)``) and joins the process group -- backend ``nccl`` (= RCCL over xGMI) when
     it has a GPU, ``gloo`` otherwise / on request (tests: two ranks sharing one GPU);
  2. calls ``allrank_amd.install(fit=True)``, which under a process group ALSO rebinds the three names through which main.py
     decides where and how wide to train:
       * ``get_torch_device`` (allrank/models/model_utils.py:13-18: always ``cuda:0``; read by main.py:71, rank_and_click.py:69,
         inference_utils.py:41 and -- as the device of freshly made tensors -- losses/neuralNDCG.py:27,93, loss_utils.py:44,98,
         bce.py:16, ordinal.py:16,34) -> this rank's device;
       * ``CustomDataParallel`` (model_utils.py:40-53; main.py:76-78 wraps the model in it when ``device_count() > 1``) -> the
         model itself: replication is what the process group does, ``score`` already exists on the model;
       * ``create_data_loaders`` (allrank/data/dataset_loading.py:230-248: "multiplying the batch size by the processing units
         count", :240-241) -> the same two loaders with the processing-unit count = the WORLD SIZE, so the global batch stays
         ``n_gpus x batch_size`` whether a rank sees all GPUs of the node or only its own (HIP_VISIBLE_DEVICES);
  3. runs ``allrank.main.run()`` as it is.  Every rank seeds identically (main.py:36-38), builds the same model and iterates the
     SAME global batches (``fit`` checks the first one with an all-gathered checksum); ``allrank_amd.fit.fit`` gives rank r its
     contiguous block of every batch (the reference's DataParallel.scatter on dim 0), the losses divide by the global batch, the
     flat gradient is all-reduced(SUM) in buckets behind the backward.  Rank 0 writes the job directory
     (results, ``model.pkl``, ``experiment_result.json``); the other ranks run main.py's bookkeeping into a private temporary
     directory that is removed at exit.
"""
import argparse
import atexit
import logging
import os
import shutil
import socket
import subprocess
import sys
import tempfile

import torch

log = logging.getLogger("allrank_amd.launch")

_state = {"device": None, "world": 1, "rank": 0, "local_rank": 0, "backend": None, "owns_group": False}


# ---------------------------------------------------------------------------------------------------------------------
# per-rank side
# ---------------------------------------------------------------------------------------------------------------------
def distributed_env(env=None):
    """(rank, local_rank, world_size) announced by a launcher (this module's ``spawn`` or torchrun), or None"""
    env = os.environ if env is None else env
    try:
        world = int(env.get("WORLD_SIZE", "1"))
    except ValueError:
        return None
    if world <= 1 or "RANK" not in env:
        return None
    rank = int(env["RANK"])
    return rank, int(env.get("LOCAL_RANK", rank)), world


def _one_gpu_per_rank_view(local_rank, n_visible):
    """True when this rank sees only ITS OWN GPU: the launcher gave every rank a private HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES
    (one visible device, several local ranks) -- the rank's GPU is then index 0 whatever its local rank."""
    try:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
    except ValueError:
        local_world = 1
    masked = any(os.environ.get(k) not in (None, "") for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"))
    return n_visible == 1 and masked and (local_world > 1 or local_rank > 0)


def _device_of(local_rank, backend, devices=None):
    """the device of a local rank: GPU ``local_rank`` of the visible set; GPU 0 when the launcher masked the set down to this rank's
    own GPU (per-rank HIP_VISIBLE_DEVICES).  ``devices`` (``--devices 0,0`` / ALLRANK_AMD_DEVICES): explicit GPU index per local rank
    -- several ranks may share one GPU under gloo (RCCL refuses duplicate devices of one shared visible set, so that is rejected
    here with a clear message)."""
    if not torch.cuda.is_available():
        return torch.device("cpu")
    n = torch.cuda.device_count()
    private = _one_gpu_per_rank_view(local_rank, n)
    if devices:
        idx = int(devices[local_rank % len(devices)])
    else:
        idx = 0 if private else local_rank
    if idx >= n:
        raise RuntimeError("allrank_amd.launch: local rank %d wants GPU %d but only %d device(s) are visible -- start at most one "
                           "rank per visible GPU, or name the device of every local rank with --devices" % (local_rank, idx, n))
    if backend == "nccl" and devices and not private and len(set(int(d) for d in devices)) < len(devices):
        raise RuntimeError("allrank_amd.launch: --devices %s puts several ranks on one GPU; RCCL needs one GPU per rank "
                           "(use --backend gloo for a shared-GPU test run)" % (",".join(str(d) for d in devices),))
    return torch.device("cuda", idx)


def setup(backend=None, devices=None, timeout_s=1800, init=True):
    """Bind this rank's GPU and join the process group the environment announces.  Idempotent; returns the rank's device.
    With no launcher environment (WORLD_SIZE unset or 1) nothing is initialised and the reference's own device rule applies.
    ``init=False`` (what a bare ``install()`` uses): never call ``init_process_group`` -- only ADOPT a group somebody
    has already initialised with more than one rank (its backend, the current device); returns None otherwise."""
    import datetime
    import torch.distributed as dist
    if _state["device"] is not None:
        return _state["device"]
    envd = distributed_env()
    backend = backend or os.environ.get("ALLRANK_AMD_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if devices is None and os.environ.get("ALLRANK_AMD_DEVICES"):
        devices = [d for d in os.environ["ALLRANK_AMD_DEVICES"].split(",") if d != ""]
    if envd is None or not init:
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # a group somebody else initialised (a host program embedding the engine): adopt it, current device as it is
            rank, world = dist.get_rank(), dist.get_world_size()
            dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
            _state.update(device=dev, world=world, rank=rank, local_rank=dev.index or 0, backend=dist.get_backend())
            return dev
        return None
    rank, local_rank, world = envd
    dev = _device_of(local_rank, backend, devices)
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl" and dev.type == "cuda":
            kw["device_id"] = dev                       # binds the communicator to this GPU (no "guessing device" barrier)
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s), **kw)
        _state["owns_group"] = True
    _state.update(device=dev, world=world, rank=rank, local_rank=local_rank, backend=backend)
    log.info("allrank_amd.launch: rank %d / %d on %s, backend %s", rank, world, dev, backend)
    return dev


def shutdown():
    import torch.distributed as dist
    if _state["owns_group"] and dist.is_initialized():
        dist.destroy_process_group()
    _state.update(device=None, world=1, rank=0, local_rank=0, backend=None, owns_group=False)


def world_size():
    return _state["world"]


def rank():
    return _state["rank"]


# -- the three names install() binds under a process group -------------------------------------------------------------
def get_torch_device():
    """this rank's device (reference: ``cuda:0`` for everybody, allrank/models/model_utils.py:13-18)"""
    dev = _state["device"]
    if dev is None:
        return torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")
    return dev


def CustomDataParallel(model, *args, **kwargs):
    """main.py:76-78 / rank_and_click.py:78-80 wrap the model when several GPUs are visible.  One process per GPU: the model of a
    rank IS its replica (``score`` and ``state_dict()`` keys unchanged -- the reference's wrapper prefixes every key with
    ``module.``, model_utils.py:40-53)."""
    return model


def create_data_loaders(train_ds, val_ds, num_workers, batch_size):
    """allrank/data/dataset_loading.py:230-248 with the number of processing units = the world size (``allrank_amd.data``'s rule):
    train loader shuffled, validation loader not, drop_last False, ``world x batch_size`` slates per GLOBAL batch.  Device-resident
    datasets (``allrank_amd.data.load_libsvm_dataset``, bound by install() when a GPU is present) get a DeviceLoader: the batch order of
    the reference's loader under the same seeds, each rank assembling only ITS block of every global batch in HBM.  Host datasets keep
    the reference's torch DataLoader: its sampler draws from torch's global generator exactly as the reference's does, so identically
    seeded ranks (main.py:36-38) iterate the same global batches and ``fit`` slices each rank's block out of them."""
    from . import data as ED
    return ED.create_data_loaders(train_ds, val_ds, num_workers, batch_size)


def _private_job_dir(argv):
    """ranks > 0: main.py's bookkeeping (output dirs, log file, used_config.json, experiment_result.json) goes to a private
    directory -- N ranks writing the same files of the real job directory would race.  Returns the rewritten argv."""
    tmp = tempfile.mkdtemp(prefix="allrank_amd_rank%d_" % _state["rank"])
    atexit.register(shutil.rmtree, tmp, ignore_errors=True)
    out, i, done = [], 0, False
    while i < len(argv):
        a = argv[i]
        if a == "--job-dir" and i + 1 < len(argv):
            out += [a, tmp]
            i += 2
            done = True
        elif a.startswith("--job-dir="):
            out.append("--job-dir=" + tmp)
            i += 1
            done = True
        else:
            out.append(a)
            i += 1
    if not done:
        out += ["--job-dir", tmp]
    return out


def run_main(main_args, backend=None, devices=None, fit=True):
    """what one rank does: setup() -> install(fit=True, distributed) -> ``allrank.main.run()`` with ``main_args`` as its command
    line.  Returns what install() rebound."""
    import importlib
    from .install import install
    dev = setup(backend, devices)
    if not fit and _state["world"] > 1:
        # the reference's own epoch loop (train_utils.py:78-147) has no gradient exchange: N ranks would train N identical replicas on the
        # whole global batch each -- N times the work for the result of one GPU
        raise RuntimeError("allrank_amd.launch: --no-fit keeps the reference's epoch loop, which cannot shard a batch over %d ranks; "
                           "drop --no-fit (allrank_amd.fit.fit shards every batch and all-reduces the gradients) or run one rank" % _state["world"])
    main = importlib.import_module("allrank.main")      # first: install() then finds main's own `from ... import` copies and records the
    done = install(fit=fit)                             # TRUE originals for uninstall()
    argv = list(main_args)
    if dev is not None and _state["rank"] > 0:
        argv = _private_job_dir(argv)
    old = sys.argv
    sys.argv = ["allrank"] + argv
    try:
        main.run()
    finally:
        sys.argv = old
    return done


# ---------------------------------------------------------------------------------------------------------------------
# launcher side
# ---------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank_, world, port, base=None, backend=None, devices=None):
    """environment of one rank of a single-node job (what torchrun would export)"""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank_), LOCAL_RANK=str(rank_), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL / tensor sharing across processes on this platform)
    if backend:
        env["ALLRANK_AMD_BACKEND"] = backend
    if devices:
        env["ALLRANK_AMD_DEVICES"] = ",".join(str(d) for d in devices)
    return env


def spawn(nproc, cmd, backend=None, devices=None, port=None, timeout=None, log_dir=None):
    """Run ``cmd`` (argv list) as ``nproc`` ranks of one node and wait.  The first rank that fails takes the others down (each by
    its own PID).  Returns the exit code (0 = every rank succeeded).  ``log_dir``: rank r's stdout+stderr go to rank<r>.log."""
    import time
    port = port or free_port()
    procs, files = [], []
    for r in range(nproc):
        fh = None
        if log_dir:
            os.makedirs(log_dir, exist_ok=True)
            fh = open(os.path.join(log_dir, "rank%d.log" % r), "w")
            files.append(fh)
        procs.append(subprocess.Popen(list(cmd), env=rank_env(r, nproc, port, backend=backend, devices=devices),
                                      stdout=fh, stderr=subprocess.STDOUT if fh else None))
    t0, code = time.time(), 0
    try:
        live = set(range(nproc))
        while live:
            for r in list(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0 and code == 0:
                    code = rc
            if code != 0 or (timeout and time.time() - t0 > timeout):
                if code == 0:
                    code = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        for fh in files:
            fh.close()
    return code


def main(argv=None):
    ap = argparse.ArgumentParser("allrank_amd.launch", description="allrank/main.py on N GPUs: one process per GPU, slate-sharded")
    ap.add_argument("--nproc", type=int, default=None, help="ranks to start (default: every visible GPU)")
    ap.add_argument("--backend", default=None, help="nccl (RCCL; default with GPUs) or gloo")
    ap.add_argument("--devices", default=None, help="GPU index per local rank, comma separated (default: rank r -> GPU r)")
    ap.add_argument("--master-port", type=int, default=None)
    ap.add_argument("--no-fit", action="store_true", help="keep the reference's epoch loop (losses / model / metrics rebound only)")
    ap.add_argument("main_args", nargs=argparse.REMAINDER, help="-- followed by the arguments of allrank/main.py")
    a = ap.parse_args(argv)
    rest = a.main_args[1:] if a.main_args[:1] == ["--"] else a.main_args
    devices = [d for d in a.devices.split(",")] if a.devices else None
    logging.basicConfig(level=logging.INFO)
    nproc = a.nproc or max(1, torch.cuda.device_count())
    if distributed_env() is not None or nproc == 1:               # one rank of a job (torchrun / our own spawn), or a one-GPU run: work
        try:
            run_main(rest, a.backend, devices, fit=not a.no_fit)
        finally:
            shutdown()
        return 0
    cmd = [sys.executable, "-m", "allrank_amd.launch"]
    if a.no_fit:
        cmd.append("--no-fit")
    cmd += ["--"] + rest
    return spawn(nproc, cmd, backend=a.backend, devices=devices, port=a.master_port)


if __name__ == "__main__":
    sys.exit(main())
