"""Dataset front ends with the reference's names (dataloaders/__init__.py)."""
from .midair import DataLoaderMidAir as MidAir
from .kitti import DataLoaderKittiRaw as KittiRaw
from .tartanair import DataLoaderTartanAir as TartanAir
from .generic import DataloaderParameters, DataLoaderGeneric, read_trajectory_csv


def get_loader(name: str):
    available = {"midair": MidAir, "kitti-raw": KittiRaw, "tartanair": TartanAir}
    try:
        return available[name]()
    except KeyError:
        print("Dataloaders available:")
        print(available.keys())
        raise NotImplementedError
