from radargnn_amd.data import get_data_loaders  # noqa: F401
