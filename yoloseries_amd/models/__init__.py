from .normal import *  # noqa: F401,F403
