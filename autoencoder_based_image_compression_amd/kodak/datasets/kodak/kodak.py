"""The Kodak test set builder of the reference (kodak_tensorflow/datasets/kodak/kodak.py) on the mirrored surface.

`create_kodak` :11-80 and `download_option` :82-106 with the reference's arguments, files and messages: the 24 RGB
pictures `kodim01.png` ... `kodim24.png` become one uint8 array (24, 512, 768) of BT.601 luminances (the three
portrait pictures turned by `numpy.rot90`, their indices kept in `list_rotation`) -- the array the compression path of
`reconstructing_eae_kodak.py` consumes. The colour conversion runs on the MI355X (`tls.rgb_to_ycbcr` ->
`eae_hip_rgb_to_ycbcr`).
"""
import os
import pickle

import numpy
import six.moves.urllib

from ...tools import tools as tls


def create_kodak(source_url, path_to_folder_rgbs, path_to_kodak, path_to_list_rotation):
    """Creates the Kodak test set (:11-80).

    Raises
    ------
    ValueError
        If a RGB image is neither 512x768x3 nor 768x512x3.
    """
    if os.path.isfile(path_to_kodak) and os.path.isfile(path_to_list_rotation):
        print('"{0}" and "{1}" already exist.'.format(path_to_kodak, path_to_list_rotation))
        print('Delete them manually to recreate the Kodak test set.')
        return
    download_option(source_url, path_to_folder_rgbs)
    (height_kodak, width_kodak) = (512, 768)
    reference_uint8 = numpy.zeros((24, height_kodak, width_kodak), dtype=numpy.uint8)
    list_rotation = []
    for i in range(24):
        path_to_file = os.path.join(path_to_folder_rgbs, 'kodim' + str(i + 1).rjust(2, '0') + '.png')
        rgb_uint8 = tls.read_image_mode(path_to_file, 'RGB')
        luminance_uint8 = tls.rgb_to_ycbcr(rgb_uint8)[:, :, 0]
        (height_image, width_image) = luminance_uint8.shape
        if height_image == height_kodak and width_image == width_kodak:
            reference_uint8[i, :, :] = luminance_uint8
        elif width_image == height_kodak and height_image == width_kodak:
            reference_uint8[i, :, :] = numpy.rot90(luminance_uint8)
            list_rotation.append(i)
        else:
            raise ValueError('"{0}" is neither {1}x{2}x3 nor {2}x{1}x3.'.format(path_to_file, height_kodak, width_kodak))
    numpy.save(path_to_kodak, reference_uint8)
    with open(path_to_list_rotation, 'wb') as file:
        pickle.dump(list_rotation, file, protocol=2)


def download_option(source_url, path_to_folder_rgbs):
    """Downloads the Kodak RGB images that are not in `path_to_folder_rgbs` yet (:82-106)."""
    for i in range(24):
        filename = 'kodim' + str(i + 1).rjust(2, '0') + '.png'
        path_to_file = os.path.join(path_to_folder_rgbs, filename)
        if os.path.isfile(path_to_file):
            print('"{}" already exists. The image is not downloaded.'.format(path_to_file))
        else:
            six.moves.urllib.request.urlretrieve(os.path.join(source_url, filename), path_to_file)
            print('Successfully downloaded "{}".'.format(filename))
