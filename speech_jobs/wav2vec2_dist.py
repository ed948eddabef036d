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


def main(argv=None):
    parser = argparse.ArgumentParser(description="Wav2Vec2 Distributed Speech Recognition")
    parser.add_argument("--num_batches", type=int, default=5, help="num_batches per replica, default is set 5")
    parser.add_argument("--batch_size", type=int, default=1, help="batch size per replica, default is set 1")
    parser.add_argument("--model_size", type=str, default="small", choices=["tiny", "small", "base"])
    parser.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    parser.add_argument("--dropout", choices=["reference", "off"], default=None,
                        help="reference = the Dropout layers of V:69-71 active (default on the bf16 path); off = parity mode")
    args = parser.parse_args(argv)

    import torch
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D
    from tethys_speech_amd import train

    task_type, task_index, rank, world = D.task_from_env()
    local_rank = int(os.environ.get("LOCAL_RANK", rank % max(1, torch.cuda.device_count())))
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    strategy = D.DataParallelStrategy(rank, world)

    workspace = os.environ.get("TETHYS_WORKSPACE", "/workspace")
    result_root = os.environ.get("TETHYS_RESULT", "/result")
    print(f"batch size per replica: {args.batch_size}, global batch size: {args.batch_size * world}")
    print(f"num_batches: {args.num_batches}")
    print(f"model_size: {args.model_size}")
    for helper in ("network.sh", "gpu.sh"):
        path = os.path.join(workspace, helper)
        if os.path.exists(path):
            os.system(f"sh {path} &")

    start_time = time.time()
    train.train_wav2vec2(strategy, model_type="pretraining", model_size=args.model_size, batch_size=args.batch_size,
                         num_batches=args.num_batches, precision=args.precision, device=device,
                         checkpoint_dir=os.path.join(workspace, "checkpoints"),
                         dropout=None if args.dropout is None else args.dropout == "reference")
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
        print(f"JCT 파일 저장 실패: {e}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
