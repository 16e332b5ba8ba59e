"""Pinhole camera intrinsics with the constructor signature of the reference's
``Camera`` (slam/common/camera.py): ``Camera(fx, fy, cx, cy, width, height)``,
positional or by keyword.  A plain class (not a dataclass) that also offers the
derived quantities the engine needs."""
import numpy as np


class Camera:
    __slots__ = ('fx', 'fy', 'cx', 'cy', 'width', 'height')

    def __init__(self, fx, fy, cx, cy, width, height):
        self.fx, self.fy = float(fx), float(fy)
        self.cx, self.cy = float(cx), float(cy)
        self.width, self.height = int(width), int(height)

    @property
    def K(self):
        """3x3 intrinsic matrix"""
        return np.array([[self.fx, 0., self.cx], [0., self.fy, self.cy],
                         [0., 0., 1.]])

    @property
    def n_pixels(self):
        return self.width * self.height

    def __eq__(self, other):
        return isinstance(other, Camera) and all(
            getattr(self, k) == getattr(other, k) for k in self.__slots__)

    def __repr__(self):
        return ('Camera(' + ', '.join(f'{k}={getattr(self, k)}'
                                      for k in self.__slots__) + ')')
