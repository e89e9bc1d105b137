"""Drop-in `basicsr` package exposing the network_g classes of the JIRF reference
(`basicsr.models.archs`) backed by libjirf_hip.so.  Only the inference forward
path is provided."""
