"""MI355X-native Jukebox sampling path: the reference's Python surface over libjukebox_hip.so (jukebox_amd/csrc)."""
