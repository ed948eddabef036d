#!/usr/bin/env python3
"""Drop-in for the reference's ``speech_jobs/wav2vec2_dist.py`` command line (V:1443-1487).

Same flags and defaults (--batch_size 1 per replica, --num_batches 5, --model_size
{tiny,small,base} default small), same stdout lines and ``<type>_<index>_jct.txt`` result
file (written inside try/except as V:1427-1435 does).  One process per GPU; cluster from
TF_CONFIG or RANK/WORLD_SIZE; roots overridable by TETHYS_WORKSPACE / TETHYS_RESULT.
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
    parser = argparse.ArgumentParser(description="Wav2Vec2 Distributed Speech Recognition")
    parser.add_argument("--num_batches", type=int, default=5, help="num_batches per replica, default is set 5")
    parser.add_argument("--batch_size", type=int, default=1, help="batch size per replica, default is set 1")
    parser.add_argument("--model_size", type=str, default="small", choices=["tiny", "small", "base"])
    parser.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    parser.add_argument("--resume_from", default=None, help="checkpoint to restore before training")
    parser.add_argument("--dropout", choices=["reference", "off"], default=None,
                        help="reference = the Dropout layers of V:69-71 active (default on the bf16 path); off = parity mode")
    args = parser.parse_args(argv)

    import torch
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D
    from tethys_speech_amd import train

    _, _, rank, world = D.task_from_env()
    import json
    task_config = json.loads(os.environ.get("TF_CONFIG") or "{}").get("task", {})
    task_type, task_index = task_config.get("type"), task_config.get("index")  # V:1453-1455: None without TF_CONFIG
    local_rank = int(os.environ.get("LOCAL_RANK", rank % max(1, torch.cuda.device_count())))
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    strategy = D.DataParallelStrategy(rank, world, backend=os.environ.get("TETHYS_DIST_BACKEND"))

    workspace = os.environ.get("TETHYS_WORKSPACE", "/workspace")
    result_root = os.environ.get("TETHYS_RESULT", "/result")
    print(f"선택된 모델 크기: {args.model_size}")
    print(f"batch size per replica: {args.batch_size}, global batch size: {args.batch_size * world}")
    print(f"num_batches: {args.num_batches}")
    print("Wav2Vec2 분산 학습 시작...")  # V:1381-1392
    print(f"선택된 모델 크기: {args.model_size}")
    print({"tiny": "Tiny 모델: 약 15-20M 파라미터", "small": "Small 모델: 약 30-40M 파라미터"}.get(args.model_size, "Base 모델: 약 95M 파라미터"))
    print("16GB V100 GPU에 최적화된 설정")
    for helper in ("network.sh", "gpu.sh"):
        path = os.path.join(workspace, helper)
        if os.path.exists(path):
            os.system(f"sh {path} &")
    print('''
========================
network profile started!
========================''')

    start_time = time.time()
    model = train.train_wav2vec2(strategy, model_type="pretraining", model_size=args.model_size, batch_size=args.batch_size,
                         num_batches=args.num_batches, precision=args.precision, device=device,
                         checkpoint_dir=os.path.join(workspace, "checkpoints"), resume_from=args.resume_from,
                         dropout=None if args.dropout is None else args.dropout == "reference",
                         model_overrides=model_overrides, **(train_kw or {}))
    jct = time.time() - start_time
    print("Training completed.")
    print("jct:", jct)
    try:  # V:1427-1435
        save_dir_name = open(os.path.join(workspace, "model.txt")).read().strip()
        out_dir = os.path.join(result_root, save_dir_name)
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"{task_type}_{task_index}_jct.txt"), "w") as f:
            f.write("%.2f" % float(jct))
    except Exception as e:  # noqa: BLE001
        print(f"JCT 파일 저장 중 오류: {e}")
    model_path = os.path.join(workspace, "model_cache", f"wav2vec2_{args.model_size}_model")  # V:1437-1440
    if rank == 0:
        os.makedirs(os.path.dirname(model_path), exist_ok=True)
        train.save_weights(model, model_path)
    print(f"{args.model_size.capitalize()} 모델이 {model_path}에 저장되었습니다.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
