#!/usr/bin/env python3
"""Drop-in for the reference's ``speech_jobs/whisper_single.py`` command line (S:1303-1324) - BASELINE config #1
read as the file is named.  Despite its name that file contains no Whisper code: it is a single-device
Wav2Vec2-base pre-training job (768 hidden / 12 layers, 5 s clips), with the older step of
S:1143-1180: roll-based negatives (S:789-839), no replica scaling, no gradient clipping, Adam(3e-5) with Keras'
default epsilon.  The other reading of config #1 ("Whisper-tiny") is ``whisper_dist.py --model_type tiny``.

Same flags and defaults (--batch_size 4, --num_batches 40), same stdout lines, result file
``/result/<job>/single_jct.txt`` written inside try/except as S:1289-1298 does.  Roots overridable by
TETHYS_WORKSPACE / TETHYS_RESULT.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main(argv=None, model_overrides=None, clip_samples=80000):
    parser = argparse.ArgumentParser(description="wav2vec2 Single GPU Speech Recognition")
    parser.add_argument("--num_batches", type=int, default=40, help="Number of batches per epoch, default is 40")
    parser.add_argument("--batch_size", type=int, default=4, help="Batch size, default is 4")
    parser.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    parser.add_argument("--dropout", choices=["reference", "off"], default=None,
                        help="reference = the model's Dropout layers active (default on the bf16 path); off = parity mode")
    args = parser.parse_args(argv)

    import torch
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import train

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    workspace = os.environ.get("TETHYS_WORKSPACE", "/workspace")
    result_root = os.environ.get("TETHYS_RESULT", "/result")

    print(f"batch size: {args.batch_size}")
    print(f"num_batches: {args.num_batches}")
    print("Wav2Vec2 단일 GPU 학습 시작...")
    path = os.path.join(workspace, "gpu.sh")  # S:1270: the utilisation sampler, if the harness provides it
    if os.path.exists(path):
        os.system(f"sh {path} &")
    print('''
========================
GPU profile started!
========================''')

    start_time = time.time()
    model = train.train_wav2vec2_single(model_type="pretraining", batch_size=args.batch_size, num_batches=args.num_batches,
                                precision=args.precision, device=device,
                                checkpoint_dir=os.path.join(workspace, "checkpoints"),
                                dropout=None if args.dropout is None else args.dropout == "reference",
                                model_overrides=model_overrides, clip_samples=clip_samples)
    jct = time.time() - start_time
    print("Training completed.")
    print("jct:", jct)
    try:  # S:1289-1298
        save_dir_name = open(os.path.join(workspace, "model.txt")).read()
        with open(os.path.join(result_root, save_dir_name.strip(), "single_jct.txt"), "w") as f:
            f.write("%.2f" % float(jct))
    except Exception:  # noqa: BLE001
        print("JCT 파일 저장 실패, 결과 디렉토리가 없을 수 있습니다.")
    model_path = os.path.join(workspace, "model_cache", "wav2vec2_model")  # S:1300-1303
    os.makedirs(os.path.dirname(model_path), exist_ok=True)
    train.save_weights(model, model_path)
    print(f"모델이 {model_path}에 저장되었습니다.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
