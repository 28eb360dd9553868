from itertools import *  # noqa: F401,F403
from itertools import zip_longest  # noqa: F401
