"""``python -m pronerf <sub-command> ...`` = ``python -m pronerf.cli <sub-command> ...``."""
import sys

from .cli import main

if __name__ == '__main__':
    main(sys.argv[1:])
