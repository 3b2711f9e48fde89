from radargnn_amd.postprocessor import PostProcessingConfiguration  # noqa: F401
