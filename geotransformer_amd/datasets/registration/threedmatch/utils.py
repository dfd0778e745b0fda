from ...threedmatch_io import *  # noqa: F401,F403
