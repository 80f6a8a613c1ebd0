"""Entry script with upstream's name: ``python train_distributed.py --cfg-filepath ...``."""
import sys

from pytorch_distributed_training_amd.cli import main

if __name__ == "__main__":
    sys.exit(main())
