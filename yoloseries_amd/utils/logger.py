"""print_config — mirror of utils/logger.py:31-45: the flat config dict (or an argparse namespace) as a two-column table."""
import pprint

__all__ = ["print_config"]


def print_config(args):
    items = args.items() if isinstance(args, dict) else vars(args).items()
    rows = [(str(k), pprint.pformat(v)) for k, v in items if not str(k).startswith("_")]
    try:
        from tabulate import tabulate
        return tabulate(rows, headers=["keys", "values"], tablefmt="fancy_grid")
    except ImportError:
        w = max([len(k) for k, _ in rows] + [4])
        return "\n".join([f"{'keys':<{w}}  values"] + [f"{k:<{w}}  {v}" for k, v in rows])
