from .coords import grid2xy, imcoordgrid, transform_coordinates
from .img import crop_borders, extract_subimages, get_coord_grid, get_imgstack, img_pad, img_resize
from .nn import (Hook, average_weights, get_downsample_factor, get_nb_classes, gpu_usage_map, mock_forward,
                 reset_bnorm, sample_weights, set_train_rng, weights_init)
from .preproc import (array2list, array2list_, check_image_dims, get_array_memsize, init_dataloaders,
                      init_fcnn_dataloaders, num_classes_from_labels, preprocess_training_image_data,
                      preprocess_training_image_data_, to_onehot, torch_format_image)

__all__ = [n for n in dir() if not n.startswith("_")]
