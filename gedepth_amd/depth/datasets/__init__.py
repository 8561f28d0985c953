from .builder import DATASETS, PIPELINES, build_dataset
from .ddad import DDADDataset
from .kitti import KITTIDataset
from .loader import SyntheticKITTI, build_dataloader, collate

__all__ = ['DATASETS', 'PIPELINES', 'build_dataset', 'KITTIDataset', 'DDADDataset', 'SyntheticKITTI', 'build_dataloader', 'collate']
