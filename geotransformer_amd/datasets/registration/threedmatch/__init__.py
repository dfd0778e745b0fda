from ...pairs import ThreeDMatchPairDataset  # noqa: F401
from ...threedmatch_io import (  # noqa: F401
    compute_registration_error,
    compute_transform_error,
    evaluate_registration_one_scene,
    read_info_file,
    read_log_file,
    write_log_file,
)
