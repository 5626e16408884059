"""Drop-in mirror of the reference's `modules/` package (taichi-dev/taichi-nerfs): same names, signatures and
return conventions, every Taichi kernel replaced by a hand-written gfx950 HIP kernel behind include/ngp_hip.h.
Put the directory that contains this package (taichi-nerfs_amd/) ahead of the reference checkout on
PYTHONPATH and the reference's train.py / gui.py import it unchanged."""
