from radargnn_amd.postprocessor import BoxSuppressor, PredictionExtractor  # noqa: F401
