from .builder import DATASETS, PIPELINES, build_dataset
from .kitti import KITTIDataset
from .loader import SyntheticKITTI, build_dataloader, collate

__all__ = ['DATASETS', 'PIPELINES', 'build_dataset', 'KITTIDataset', 'SyntheticKITTI', 'build_dataloader', 'collate']
