# Drop-in module name for the reference's `import environment` / `from environment import ...`:
# put neuralnj_amd/compat on sys.path.
from neuralnj_amd.environment import *  # noqa: F401,F403
from neuralnj_amd import environment as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
