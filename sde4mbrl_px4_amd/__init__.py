"""MI355X-native MPC inner loop for sde4mbrl_px4 (neural-SDE rollout + APG trajectory optimiser).

Drop-in for the solver the reference obtains at
``sde4mbrl_px4/mpc_controller/sde_control.py:685`` (``load_mpc_from_cfgfile``) and calls per control
tick at ``sde_control.py:345-350,400-416`` (``m_reset`` / ``m_mpc``). The arithmetic runs in
hand-written HIP kernels (``csrc/``) behind the C ABI declared in ``include/sdempc.h``; there is no
CPU fallback: importing the solver without the built extension raises.
"""
from .config import MPCConfig, load_mpc_config
from .model import RotorSDEModel, synthetic_iris, synthetic_hexa, synthetic_multirotor

__all__ = ["MPCConfig", "load_mpc_config", "RotorSDEModel", "synthetic_iris", "synthetic_hexa", "synthetic_multirotor"]
