"""Weight I/O helpers: the a-trous decimation of VGG-16's fc6/fc7 that the reference
performs while building its graph (ssdvgg.py:245-253, 273-280), as an offline converter producing the
`vgg16_ssd.npz` that SSDVGG.build_from_vgg(vgg_dir, ...) loads, keyed by the reference's variable names."""
import numpy as np

VGG_CONVS = ['conv1_1', 'conv1_2', 'conv2_1', 'conv2_2', 'conv3_1', 'conv3_2', 'conv3_3', 'conv4_1', 'conv4_2', 'conv4_3',
             'conv5_1', 'conv5_2', 'conv5_3']


def decimate_fc6(fc6_w, fc6_b):
    """fc6 [7,7,512,4096] -> mod_conv6 [3,3,512,1024]: every 3rd tap, every 4th output (ssdvgg.py:246-253)."""
    fc6_w = np.asarray(fc6_w); fc6_b = np.asarray(fc6_b)
    if fc6_w.shape != (7, 7, 512, 4096) or fc6_b.shape != (4096,):
        raise ValueError(f'fc6 must be [7,7,512,4096] / [4096], got {fc6_w.shape} / {fc6_b.shape}')
    return np.ascontiguousarray(fc6_w[0:7:3, 0:7:3, :, 0:4096:4], np.float32), np.ascontiguousarray(fc6_b[0:4096:4], np.float32)


def decimate_fc7(fc7_w, fc7_b):
    """fc7 [1,1,4096,4096] -> mod_conv7 [1,1,1024,1024]: every 4th input and output (ssdvgg.py:274-280)."""
    fc7_w = np.asarray(fc7_w); fc7_b = np.asarray(fc7_b)
    if fc7_w.shape != (1, 1, 4096, 4096) or fc7_b.shape != (4096,):
        raise ValueError(f'fc7 must be [1,1,4096,4096] / [4096], got {fc7_w.shape} / {fc7_b.shape}')
    return np.ascontiguousarray(fc7_w[:, :, 0:4096:4, 0:4096:4], np.float32), np.ascontiguousarray(fc7_b[0:4096:4], np.float32)


def vgg16_to_ssd(vgg):
    """{'conv1_1/filter': ..., 'conv1_1/biases': ..., ..., 'fc6/weights', 'fc6/biases', 'fc7/weights',
    'fc7/biases'} (the tensors ssdvgg.py:192-207 reads from the SavedModel, filters HWIO) ->
    {'conv*/filter|biases', 'mod_conv6|7/filter|biases'} ready for SSDVGG.load_variables."""
    out = {}
    for n in VGG_CONVS:
        out[n + '/filter'] = np.ascontiguousarray(vgg[n + '/filter'], np.float32)
        out[n + '/biases'] = np.ascontiguousarray(vgg[n + '/biases'], np.float32)
    out['mod_conv6/filter'], out['mod_conv6/biases'] = decimate_fc6(vgg['fc6/weights'], vgg['fc6/biases'])
    out['mod_conv7/filter'], out['mod_conv7/biases'] = decimate_fc7(vgg['fc7/weights'], vgg['fc7/biases'])
    return out


def save_vgg_npz(path, vgg):
    np.savez(path, **vgg16_to_ssd(vgg))
