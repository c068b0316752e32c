# Drop-in module name for the reference's `import utils` / `from utils import ...`:
# put neuralnj_amd/compat on sys.path.
from neuralnj_amd.utils import *  # noqa: F401,F403
from neuralnj_amd import utils as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
