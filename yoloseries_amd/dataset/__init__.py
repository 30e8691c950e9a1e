"""Data formats either side of the hot path — mirror of the reference's dataset/ surface that the drivers use
(SURVEY §8f rank 3): collate functions, the letterbox geometry they apply, the CUDA-stream prefetcher, and a
synthetic detection dataset that produces items in the reference's __getitem__ format."""
from .data_collater import fixed_imgsize_collate_fn, normal_normalization, test_dataset_collate_fn
from .data_prefetcher import DataPrefetcher, TestDataPrefetcher
from .synthetic import SyntheticDetectionDataset
from .data_loader import build_dataloader, build_test_dataloader, build_val_dataloader

__all__ = ['fixed_imgsize_collate_fn', 'test_dataset_collate_fn', 'normal_normalization', 'DataPrefetcher',
           'TestDataPrefetcher', 'SyntheticDetectionDataset', 'build_dataloader', 'build_val_dataloader', 'build_test_dataloader']
