"""Dataset surface of the reference (ref/modules/loader.py:8-88), host-side data prep only.

Same classes, same `__getitem__` results: image -> RGB -> 256x256 -> float CHW in [0,1]
(`torchvision.transforms.ToTensor` semantics, re-implemented with numpy because torchvision is not
required here); MSCOCO: first caption per image with the reference's fixed prompt (typo included);
RedCaps: word-level span masking, 15 % of the words + 1, `<extra_id_k>` sentinels
(ref/modules/loader.py:56-72).  The COCO annotation index is read directly from the JSON
(the reference's SilentCOCO, ref/modules/coco.py, is pycocotools with the prints removed).
"""
import json
import os

import numpy as np
import torch
from PIL import Image

COCO_PROMPT = 'What does th image describe ?'  # verbatim, ref/modules/loader.py:38


def pil_to_tensor01(image):
    """ToTensor(): HWC uint8 -> CHW float32 / 255."""
    a = np.asarray(image, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(a.transpose(2, 0, 1).copy()).to(torch.float32).div_(255.0)


def span_mask(caption: str):
    """(src_text, tgt_text) of RedCapsDatasetLoader.__getitem__; consumes torch's global RNG exactly as the
    reference does (one `torch.randperm(len(words))`)."""
    for ch in '.,!?':
        caption = caption.replace(ch, ' ' + ch)
    words = caption.split()
    n_mask = int(len(words) * 0.15) + 1
    masked = set(torch.randperm(len(words))[:n_mask].tolist())
    tgt = ['<extra_id_0>']
    k = 0
    for i, w in enumerate(words):
        if i in masked:
            tgt.extend([w, f'<extra_id_{k + 1}>'])
            words[i] = f'<extra_id_{k}>'
            k += 1
    return ' '.join(words), ' '.join(tgt)


class DatasetLoader(torch.utils.data.Dataset):
    # decode_only (klab extension): __getitem__ returns the decoded HWC uint8 RGB image at its own size; the
    # 256x256 resize, ToTensor and the image processor then run on the GPU (`modules.image_pipeline.GpuImageProcessor
    # .from_decoded`, bit-identical to Pillow).  Batches of such items need `collate_decoded` (images stay a list).
    decode_only = False
    # file_bytes (klab extension): __getitem__ returns the JPEG file's bytes; decoding too then runs through
    # `GpuImageProcessor.from_jpeg` (host Huffman stage + device reconstruction, byte-identical to `Image.open().convert('RGB')`)
    file_bytes = False

    def __init__(self):
        self.images, self.tgt_texts, self.src_texts = [], [], []
        self.transform = pil_to_tensor01

    def _load_image(self, path):
        if self.file_bytes:
            with open(path, 'rb') as f:
                return f.read()
        if self.decode_only:
            return torch.from_numpy(np.asarray(Image.open(path).convert('RGB'), dtype=np.uint8).copy())
        return self.transform(Image.open(path).convert('RGB').resize((256, 256)))

    def __getitem__(self, idx):
        return self._load_image(self.images[idx]), self.src_texts[idx], self.tgt_texts[idx]

    def __len__(self):
        return len(self.images)


class COCODatasetLoader(DatasetLoader):
    def __init__(self, data_dir='/data/datatset/mscoco2017', phase='train'):
        super().__init__()
        with open(os.path.join(data_dir, 'annotations', f'captions_{phase}2017.json')) as f:
            ann = json.load(f)
        first_caption = {}
        for a in ann['annotations']:  # first annotation per image, in file order (= coco.getAnnIds(image_id)[0])
            first_caption.setdefault(a['image_id'], a['caption'])
        img_dir = os.path.join(data_dir, f'{phase}2017')
        for info in ann['images']:
            self.images.append(os.path.join(img_dir, info['file_name']))
            self.src_texts.append(COCO_PROMPT)
            self.tgt_texts.append(first_caption[info['id']])


class RedCapsDatasetLoader(DatasetLoader):
    def __init__(self, data_dir='/data/dataset/redcaps', phase='train'):
        super().__init__()
        anno_dir = os.path.join(data_dir, 'annotations')
        img_dir = os.path.join(data_dir, 'images')
        for name in os.listdir(anno_dir):
            with open(os.path.join(anno_dir, name)) as f:
                for ann in json.load(f)["annotations"]:
                    self.images.append(os.path.join(img_dir, ann["subreddit"], f"{ann['image_id']}.jpg"))
                    self.src_texts.append(ann['raw_caption'])

    def __getitem__(self, idx):
        src, tgt = span_mask(self.src_texts[idx])  # RNG is consumed before the image is opened, as in the reference
        return self._load_image(self.images[idx]), src, tgt


def collate_decoded(batch):
    """collate_fn for `decode_only` / `file_bytes` datasets: (list of HWC uint8 tensors or of JPEG byte strings, list of src texts,
    list of tgt texts)"""
    images, src, tgt = zip(*batch)
    return list(images), list(src), list(tgt)


def get_dataloader(args, phase, rank):
    name = args.data_dir.lower()
    if 'mscoco' in name:
        dataset = COCODatasetLoader(args.data_dir, phase)
    elif 'redcaps' in name:
        dataset = RedCapsDatasetLoader(args.data_dir, phase)
    else:
        raise NotImplementedError
    # the reference sizes the sampler by the LOCAL device count and never calls set_epoch: kept
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=torch.cuda.device_count(), rank=rank,
                                                              shuffle=True, drop_last=True)
    gpu_pre = bool(getattr(args, 'gpu_preprocess', False))  # klab extension; the reference has no such flag (default off)
    dataset.decode_only = gpu_pre
    dataset.file_bytes = gpu_pre and bool(getattr(args, 'gpu_jpeg_decode', False))  # klab extension: batches carry file bytes
    return torch.utils.data.DataLoader(dataset, batch_size=args.batch_size, num_workers=os.cpu_count() // 4, pin_memory=not gpu_pre,
                                       sampler=sampler, collate_fn=collate_decoded if gpu_pre else None)
