#!/usr/bin/env python3
"""Drop-in for the reference's ``stable_jobs/wav2vec2_dist.py`` command line ("T:", T:1311-1339).

The model and the step are those of ``speech_jobs/whisper_single.py`` (Wav2Vec2-base, 5 s clips, roll-based negatives,
no replica scaling, no clipping, Adam(3e-5) with Keras' default epsilon) run under MultiWorkerMirroredStrategy: gradient
SUM over replicas, summed loss (T:1143-1190).  Same flags and defaults (--batch_size 1 per replica, --num_batches 40), same
stdout lines, result file ``/result/<job>/<type>_<index>_jct.txt`` (T:1297-1303; written without a try/except there, so a
missing job directory is an error here too).  One process per GPU; cluster from TF_CONFIG or RANK/WORLD_SIZE; roots
overridable by TETHYS_WORKSPACE / TETHYS_RESULT.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main(argv=None, model_overrides=None, train_kw=None):
    """``model_overrides`` / ``train_kw`` are for tests (tiny dimensions, short clips); the command line has neither."""
    parser = argparse.ArgumentParser(description="wav2vec2 Distributed Speech Recognition")
    parser.add_argument("--num_batches", type=int, default=40, help="num_batches per replica, default is set 40")
    parser.add_argument("--batch_size", type=int, default=1, help="batch size per replica, default is set 1")
    parser.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    parser.add_argument("--dropout", choices=["reference", "off"], default=None,
                        help="reference = the model's Dropout layers active (default on the bf16 path); off = parity mode")
    args = parser.parse_args(argv)

    import torch
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D
    from tethys_speech_amd import train

    _, _, rank, world = D.task_from_env()
    task_config = json.loads(os.environ.get("TF_CONFIG") or "{}").get("task", {})
    task_type, task_index = task_config.get("type"), task_config.get("index")  # T:1319-1322: None without TF_CONFIG
    local_rank = int(os.environ.get("LOCAL_RANK", rank % max(1, torch.cuda.device_count())))
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    strategy = D.DataParallelStrategy(rank, world, backend=os.environ.get("TETHYS_DIST_BACKEND"))

    workspace = os.environ.get("TETHYS_WORKSPACE", "/workspace")
    result_root = os.environ.get("TETHYS_RESULT", "/result")
    print(f"batch size per replica: {args.batch_size}, global batch size: {args.batch_size * world}")
    print(f"num_batches: {args.num_batches}")
    print("Wav2Vec2 분산 학습 시작...")  # T:1272
    for helper in ("network.sh", "gpu.sh"):
        path = os.path.join(workspace, helper)
        if os.path.exists(path):
            os.system(f"sh {path} &")
    print('''
========================
network profile started!
========================''')

    start_time = time.time()
    model = train.train_wav2vec2_stable(strategy, model_type="pretraining", batch_size=args.batch_size,
                                        num_batches=args.num_batches, precision=args.precision, device=device,
                                        checkpoint_dir=os.path.join(workspace, "checkpoints"),
                                        dropout=None if args.dropout is None else args.dropout == "reference",
                                        model_overrides=model_overrides, **(train_kw or {}))
    jct = time.time() - start_time
    print("Training completed.")
    print("jct:", jct)
    save_dir_name = open(os.path.join(workspace, "model.txt")).read()  # T:1297-1303
    with open(os.path.join(result_root, save_dir_name.strip(), f"{task_type}_{task_index}_jct.txt"), "w") as f:
        f.write("%.2f" % float(jct))
    model_path = os.path.join(workspace, "model_cache", "wav2vec2_model")  # T:1306-1308
    if rank == 0:
        os.makedirs(os.path.dirname(model_path), exist_ok=True)
        train.save_weights(model, model_path)
    print(f"모델이 {model_path}에 저장되었습니다.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
