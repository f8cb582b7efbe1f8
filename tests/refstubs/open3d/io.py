import numpy as np
from PIL import Image


def read_image(path):
    return np.array(Image.open(path))
