from radargnn_amd.postprocessor import BoxSuppressor, Postprocessor, PredictionExtractor  # noqa: F401
