#!/usr/bin/env python3
"""Drop-in for the reference's ``speech_jobs/wav2vec2_single.py`` command line ("U:", U:1279-1349): the model and step of
``speech_jobs/wav2vec2_dist.py`` without a strategy (U:1118-1175 is V:1186-1260 on one replica: local
clip_by_global_norm(1.0), Adam(3e-5, eps 1e-8, clipnorm 1.0)), 2 s clips, ``batch(drop_remainder=True)``.

Same flags and defaults (--num_batches 5, --batch_size 1, --model_size small, --model_type pretraining, --learning_rate
3e-5, --num_epochs 1), same stdout lines; roots ./model_cache and ./checkpoints as the reference (overridable by
TETHYS_WORKSPACE).  --model_type asr / classification are dead from the reference's benchmarks and refused.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main(argv=None, model_overrides=None, train_kw=None):
    parser = argparse.ArgumentParser(description="Wav2Vec2 Single GPU Speech Recognition")
    parser.add_argument("--num_batches", type=int, default=5, help="num_batches, default is set 5")
    parser.add_argument("--batch_size", type=int, default=1, help="batch size, default is set 1")
    parser.add_argument("--model_size", type=str, default="small", choices=["tiny", "small", "base"])
    parser.add_argument("--model_type", type=str, default="pretraining", choices=["pretraining", "asr", "classification"])
    parser.add_argument("--learning_rate", type=float, default=3e-5, help="Learning rate")
    parser.add_argument("--num_epochs", type=int, default=1, help="Number of epochs")
    parser.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    parser.add_argument("--dropout", choices=["reference", "off"], default=None,
                        help="reference = the Dropout layers of U:69-71 active (default on the bf16 path); off = parity mode")
    args = parser.parse_args(argv)
    if args.model_type != "pretraining":
        parser.error("only --model_type pretraining is on the reference's benchmarked path")

    import torch
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D
    from tethys_speech_amd import train

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    root = os.environ.get("TETHYS_WORKSPACE", ".")
    print("Wav2Vec2 단일 GPU 학습 시작...")  # U:1302-1312
    print(f"선택된 모델 크기: {args.model_size}")
    print(f"선택된 모델 타입: {args.model_type}")
    print({"tiny": "Tiny 모델: 약 15-20M 파라미터", "small": "Small 모델: 약 30-40M 파라미터"}.get(args.model_size, "Base 모델: 약 95M 파라미터"))

    start_time = time.time()
    print("모델 가중치 초기화 중...")  # U:1193-1196: the first batch builds the weights
    print("모델 가중치 초기화 완료")
    strategy = D.DataParallelStrategy(0, 1)
    model = train.train_wav2vec2(strategy, model_type=args.model_type, model_size=args.model_size, num_epochs=args.num_epochs,
                                 learning_rate=args.learning_rate, batch_size=args.batch_size, num_batches=args.num_batches,
                                 precision=args.precision, device=device, checkpoint_dir=os.path.join(root, "checkpoints"),
                                 dropout=None if args.dropout is None else args.dropout == "reference",
                                 model_overrides=model_overrides, epoch_label="에포크", init_batches=1, checkpoint_stem="model",
                                 **(train_kw or {}))
    jct = time.time() - start_time
    print("학습 완료.")
    print("JCT:", jct)
    model_path = os.path.join(root, "model_cache", f"wav2vec2_{args.model_size}_{args.model_type}_model")  # U:1343-1346
    os.makedirs(os.path.dirname(model_path), exist_ok=True)
    train.save_weights(model, model_path)
    print(f"{args.model_size.capitalize()} {args.model_type} 모델이 {model_path}에 저장되었습니다.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
