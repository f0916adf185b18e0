"""sea_current_amd -- host-side Python plumbing over libsea_current_hip.so (placeholder, filled below)."""
