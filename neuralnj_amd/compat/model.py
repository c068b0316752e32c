# Drop-in module name for the reference's `import model` / `from model import ...`:
# put neuralnj_amd/compat on sys.path.
from neuralnj_amd.model import *  # noqa: F401,F403
from neuralnj_amd import model as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
