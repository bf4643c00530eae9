"""`python experiments/generate.py ...` == `python -m tinyedm.generate ...` (reference: src/tinyedm/generate.py; same
flags: --ckpt_path --load_ema --output_dir --num_samples --image_size --num_classes --batch_size --num_steps)."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for multi-process GPU runs
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tinyedm.generate import main  # noqa: E402

if __name__ == "__main__":
    main()
