"""spotlight_amd -- MI355X-native drop-in for the implicit-feedback training path of
maciejkula/spotlight (spotlight.factorization.implicit.ImplicitFactorizationModel
fit()/predict() over spotlight.interactions.Interactions).

Python here is host glue only: numpy containers, torch device memory and streams, and a
ctypes binding (spotlight_amd._native) onto csrc/libspotlight_hip.so, whose hand-written
gfx950 kernels do the sampling, gather, dot, loss, backward and optimizer update.  There is
no CPU path: without the HIP library and a GPU, fit()/predict() raise.
"""
__version__ = '0.1.0'
