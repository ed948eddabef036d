#!/usr/bin/env python3
"""Drop-in for the reference's ``speech_jobs/whisper_dist.py`` command line (W:1029-1058).

Same flags (--batch_size per replica, --num_batches), same stdout lines (banner, per-step
``Step i, Loss: ...`` line, ``Training completed.`` / ``jct:``), same result file
``/result/<job>/<task_type>_<task_index>_jct.txt`` with the job name read from
``/workspace/model.txt``.  One process per GPU; the cluster comes from TF_CONFIG (as the
TFJob harness sets it) or from RANK/WORLD_SIZE (torchrun).  Roots are overridable by
TETHYS_WORKSPACE / TETHYS_RESULT so it also runs outside the pod.  Extra, optional flags
(--precision, --model_type) default to the reference's behaviour (model "small").
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main(argv=None, model_overrides=None, train_kw=None):
    """``model_overrides`` / ``train_kw`` are for tests (tiny dimensions, short clips); the command line has neither."""
    parser = argparse.ArgumentParser(description="Whisper-small Distributed Speech Recognition")
    parser.add_argument("--num_batches", type=int, default=40, help="num_batches per replica, default is set 40")
    parser.add_argument("--batch_size", type=int, default=1, help="batch size per replica, default is set 1")
    parser.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    parser.add_argument("--model_type", default="small")
    parser.add_argument("--dropout", choices=["reference", "off"], default=None,
                        help="reference = the Dropout layers of W:29-30 active (default on the bf16 path); off = parity mode")
    parser.add_argument("--tensor_logs", default=None,
                        help="directory for the tensor-size / skewness report (whisper_dist_tensorsize.py's files)")
    parser.add_argument("--resume_from", default=None, help="checkpoint to restore before training")
    args = parser.parse_args(argv)

    import torch
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D
    from tethys_speech_amd import train

    _, _, rank, world = D.task_from_env()
    # the result-file name uses the raw TF_CONFIG task fields, as W:1037-1040 does: without TF_CONFIG they are None
    import json
    task_config = json.loads(os.environ.get("TF_CONFIG") or "{}").get("task", {})
    task_type, task_index = task_config.get("type"), task_config.get("index")
    local_rank = int(os.environ.get("LOCAL_RANK", rank % max(1, torch.cuda.device_count())))
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    strategy = D.DataParallelStrategy(rank, world, backend=os.environ.get("TETHYS_DIST_BACKEND"))

    workspace = os.environ.get("TETHYS_WORKSPACE", "/workspace")
    result_root = os.environ.get("TETHYS_RESULT", "/result")
    global_batch = args.batch_size * strategy.num_replicas_in_sync
    print(f"batch size per replica: {args.batch_size}, global batch size: {global_batch}")
    print(f"num_batches: {args.num_batches}")
    print("Whisper-small 분산 학습 시작...")
    for helper in ("network.sh", "gpu.sh"):  # W:994-995: side samplers, only if the harness provides them
        path = os.path.join(workspace, helper)
        if os.path.exists(path):
            os.system(f"sh {path} &")
    print('''
========================
network profile started!
========================''')

    start_time = time.time()
    model = train.train_whisper(strategy, model_type=args.model_type, batch_size=args.batch_size,
                                num_batches=args.num_batches, precision=args.precision, device=device,
                                checkpoint_dir=os.path.join(workspace, "checkpoints"),
                                tensor_log_dir=args.tensor_logs, resume_from=args.resume_from,
                                dropout=None if args.dropout is None else args.dropout == "reference",
                                model_overrides=model_overrides, **(train_kw or {}))
    jct = time.time() - start_time
    print("Training completed.")
    print("jct:", jct)

    model_txt = os.path.join(workspace, "model.txt")
    if os.path.exists(model_txt):
        save_dir_name = open(model_txt).read().strip()
        out_dir = os.path.join(result_root, save_dir_name)
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"{task_type}_{task_index}_jct.txt"), "w") as f:
            f.write("%.2f" % float(jct))
    elif workspace == "/workspace":
        # W:1016 has no guard: a missing model.txt is an uncaught exception -> non-zero exit
        raise FileNotFoundError(model_txt)
    # W:1024-1026: model.save_weights(CACHE_DIR/whisper_small_model)
    model_path = os.path.join(workspace, "model_cache", "whisper_small_model")
    if rank == 0:
        os.makedirs(os.path.dirname(model_path), exist_ok=True)
        train.save_weights(model, model_path)
    print(f"모델이 {model_path}에 저장되었습니다.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
