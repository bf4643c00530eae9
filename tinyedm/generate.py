"""`python -m tinyedm.generate ...` (reference: src/tinyedm/generate.py): the implementation lives in tinyedm_amd.generate."""
from tinyedm_amd.generate import CIFAR_MEAN, CIFAR_STD, generate, main  # noqa: F401

if __name__ == "__main__":
    main()
