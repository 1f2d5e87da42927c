"""MI355X-native hot path of mobeets/classifying-vae-lstm (cl_vae / cl_vrnn).

Layout mirrors the reference's ``code/`` directory for the path it replaces:
``cl_vae/{model,train,sample}.py``, ``cl_vrnn/{model,train,sample}.py`` and the
``utils/`` they import; underneath, every tensor op is a hand-written HIP kernel
in ``csrc/`` reached through the C ABI of ``include/clvae.h`` (``_lib.py``).
There is no CPU fallback: without the HIP library and a gfx950 device the
compute entry points raise.
"""
__version__ = "0.1.0"
