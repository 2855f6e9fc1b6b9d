"""Help formatting for the command line (cosmetic; counterpart of TrueConsense/func.py)."""
import argparse


class MyHelpFormatter(argparse.RawTextHelpFormatter):
    def __init__(self, prog):
        super().__init__(prog, max_help_position=40, width=100)


class color:
    YELLOW = "\033[93m"
    END = "\033[0m"
