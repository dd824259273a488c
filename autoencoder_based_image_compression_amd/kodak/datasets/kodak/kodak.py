"""The Kodak test-set builder with the reference's call surface (kodak_tensorflow/datasets/kodak/kodak.py: `create_kodak`
:11-80, `download_option` :82-106; same arguments, files and messages).

The 24 RGB pictures `kodim01.png` ... `kodim24.png` become one uint8 array (24, 512, 768) of BT.601 luminances, the three
portrait pictures turned onto their side and their indices kept in `list_rotation`: the array the compression path of
`reconstructing_eae_kodak.py` consumes. The colour conversion runs on the MI355X (`tls.rgb_to_ycbcr` ->
`eae_hip_rgb_to_ycbcr`).
"""
import os
import pickle

import numpy
import six.moves.urllib

from ...tools import tools as tls

NB_PICTURES = 24
LANDSCAPE = (512, 768)


def _picture_name(index):
    return 'kodim{:02d}.png'.format(index + 1)


def create_kodak(source_url, path_to_folder_rgbs, path_to_kodak, path_to_list_rotation):
    """Creates the Kodak test set: `path_to_kodak` (".npy") and `path_to_list_rotation` (".pkl").

    Raises
    ------
    ValueError
        If a RGB image is neither 512x768x3 nor 768x512x3.
    """
    if all(os.path.isfile(path) for path in (path_to_kodak, path_to_list_rotation)):
        print('"{0}" and "{1}" already exist.'.format(path_to_kodak, path_to_list_rotation))
        print('Delete them manually to recreate the Kodak test set.')
        return
    download_option(source_url, path_to_folder_rgbs)
    luminances = numpy.zeros((NB_PICTURES,) + LANDSCAPE, dtype=numpy.uint8)
    turned = []
    for index in range(NB_PICTURES):
        path = os.path.join(path_to_folder_rgbs, _picture_name(index))
        luminance = tls.rgb_to_ycbcr(tls.read_image_mode(path, 'RGB'))[:, :, 0]
        if luminance.shape == LANDSCAPE:
            luminances[index] = luminance
        elif luminance.shape == LANDSCAPE[::-1]:
            luminances[index] = numpy.rot90(luminance)          # portrait: a quarter turn, undone when pictures are written
            turned.append(index)
        else:
            raise ValueError('"{0}" is neither {1}x{2}x3 nor {2}x{1}x3.'.format(path, LANDSCAPE[0], LANDSCAPE[1]))
    numpy.save(path_to_kodak, luminances)
    with open(path_to_list_rotation, 'wb') as file:
        pickle.dump(turned, file, protocol=2)


def download_option(source_url, path_to_folder_rgbs):
    """Fetches the pictures that are not in `path_to_folder_rgbs` yet from `source_url`."""
    for name in map(_picture_name, range(NB_PICTURES)):
        target = os.path.join(path_to_folder_rgbs, name)
        if os.path.isfile(target):
            print('"{}" already exists. The image is not downloaded.'.format(target))
            continue
        six.moves.urllib.request.urlretrieve(os.path.join(source_url, name), target)
        print('Successfully downloaded "{}".'.format(name))
