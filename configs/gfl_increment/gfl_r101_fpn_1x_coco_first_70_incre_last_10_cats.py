# Stage 2 of the 70+10 protocol (BASELINE.json configs[3]): an 80-class R101 student learns the last 10
# categories; ERD distils the 70 old-class responses of the frozen stage-1 teacher.  Derived from the R50 40+40
# config; not a file of the reference.
_base_ = './gfl_r50_fpn_1x_coco_first_40_incre_last_40_cats.py'

model = dict(
    ori_setting=dict(
        ori_checkpoint_file='../ERD_results/gfl_increment/gfl_r101_fpn_1x_coco_first_70_cats/epoch_12.pth',
        ori_num_classes=70,
        ori_config_file='configs/gfl_increment/gfl_r101_fpn_1x_coco_first_70_cats.py'),
    backbone=dict(depth=101, init_cfg=dict(type='Pretrained', checkpoint='torchvision://resnet101')))

train_dataloader = dict(dataset=dict(ann_file='annotations/instances_train2017_sel_last_10_cats.json'))
