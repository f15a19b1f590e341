"""History buffer of generated images for the discriminator (reference:
ctu/utils/image_pool.py:12-39).  The JPD-SE scripts train with pool_size == 0, where the pool is
the identity; a non-empty pool mixes samples across steps, which breaks the per-image
independence that data parallelism relies on, so it is refused here exactly where the
reference refuses it for multi-GPU (pix2pixHD_model.py:199-200)."""


class ImagePool(object):

  def __init__(self, pool_size):
    if pool_size != 0:
      raise NotImplementedError('Fake Pool (pool_size > 0) is not implemented on the HIP data-parallel path')
    self.pool_size = 0

  def query(self, images):
    return images
