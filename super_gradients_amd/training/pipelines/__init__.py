from .pipelines import DetectionPipeline, Pipeline  # noqa: F401
