"""Host-side mirror of /root/reference/supervision/ (the loss of the depth training scripts)."""
