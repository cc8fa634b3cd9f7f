"""paradis_model_amd: MI355X-native (gfx950) implementation of the PARADIS
advection-diffusion-reaction forward/backward hot path behind the reference's
``model.*`` nn.Module API.  Compute runs in hand-written HIP kernels exposed by
the C-ABI library ``libparadis_hip.so`` (see ``include/paradis_hip.h``)."""

__version__ = "0.1.0"
