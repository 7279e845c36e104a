"""Minimal holder for real 2-D data with its sampling (prysm/_richdata.py is out of scope:
slicing, interpolation and plotting are host-side conveniences).  Keeps the three attributes the
propagation path reads: ``data`` (device tensor), ``dx``, ``wavelength``.
"""
from .mathops import array_to_true_numpy


class RichData:
    """Wrapper of a data array, its inter-sample spacing and the wavelength (prysm/_richdata.py RichData)."""

    def __init__(self, data, dx, wavelength):
        self.data = data
        self.dx = dx
        self.wavelength = wavelength

    @property
    def shape(self):
        return tuple(self.data.shape)

    @property
    def size(self):
        return self.data.numel() if hasattr(self.data, 'numel') else self.data.size

    def numpy(self):
        """The data as a true numpy array (host copy)."""
        return array_to_true_numpy(self.data)

    def copy(self):
        return RichData(self.data.clone() if hasattr(self.data, 'clone') else self.data.copy(), self.dx, self.wavelength)
