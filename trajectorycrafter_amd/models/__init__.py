"""Host-side mirror of the reference's `models/` package (crosstransformer3d, autoencoder_magvit,
pipeline_trajectorycrafter) backed by libtcx_hip.so."""
