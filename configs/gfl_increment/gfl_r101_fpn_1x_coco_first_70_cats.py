# Stage 1 of the 70+10 protocol on a ResNet-101 backbone (BASELINE.json configs[3]).  Not a file of the
# reference (it ships the R50 40+40 pair only): the R50 first-40 config with a deeper backbone and 70 classes.
_base_ = './gfl_r50_fpn_1x_coco_first_40_cats.py'

data_root = '../data/coco/'

model = dict(
    backbone=dict(depth=101, init_cfg=dict(type='Pretrained', checkpoint='torchvision://resnet101')),
    bbox_head=dict(num_classes=70))

train_dataloader = dict(dataset=dict(ann_file='annotations/instances_train2017_sel_first_70_cats.json'))
val_dataloader = dict(dataset=dict(ann_file='annotations/instances_val2017_sel_first_70_cats.json'))
val_evaluator = dict(ann_file=data_root + 'annotations/instances_val2017_sel_first_70_cats.json')
test_dataloader = val_dataloader
test_evaluator = val_evaluator
