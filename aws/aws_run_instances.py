#!/usr/bin/env python
"""EC2-style provisioning helper (reference: aws/aws_run_instances.py:4-61): look up the machine image, start the
actor fleet and the learner host, wait until they run and print the fab commands. boto3 is optional and only
imported when the script actually talks to a cloud; `--dry-run` prints the plan from the YAML alone."""
import argparse

import yaml


def load_config(path):
    with open(path) as fh:
        return yaml.safe_load(fh)          # the reference uses yaml.load (unsafe); safe_load here


def plan(config):
    n_act = config['num_actors_instance']
    per = config['num_actors_process_for_each_instance']
    return dict(actor_instances=n_act, actors_total=n_act * per, actor_type=config['actors_instance_type'],
                learner_type=config['learner_instance_type'], image=config.get('image_name', 'distributed_rl_amd'))


def fab_commands(config, actors_ip, learner_ip):
    key = config['ssh_key_file']
    return ("fab -H %s -u ubuntu -i %s learner_run" % (learner_ip, key),
            "fab -H %s -u ubuntu -i %s actor_run:num_proc=%d,learner_host=%s"
            % (','.join(actors_ip), key, config['num_actors_process_for_each_instance'], learner_ip))


def run_instances_and_wait(config):
    import boto3  # optional dependency
    ec2 = boto3.client('ec2')
    image_id = ec2.describe_images(Owners=["self"], Filters=[{'Name': 'name', 'Values': [config.get('image_name', 'distributed_rl_amd')]}])['Images'][0]['ImageId']

    def start(instance_type, count=1):
        r = ec2.run_instances(ImageId=image_id, InstanceType=instance_type, KeyName=config['key_name'], MaxCount=count, MinCount=count)
        return [i['InstanceId'] for i in r['Instances']]

    def ips(ids):
        res = ec2.describe_instances(InstanceIds=ids)
        return [i['PublicIpAddress'] for rsv in res['Reservations'] for i in rsv['Instances']]

    act = start(config['actors_instance_type'], config['num_actors_instance'])
    lrn = start(config['learner_instance_type'])
    print('Wait until instances running...')
    ec2.get_waiter('instance_running').wait(InstanceIds=act + lrn)
    print("Success in running all instances.")
    lcmd, acmd = fab_commands(config, ips(act), ips(lrn)[0])
    print("Learner's command: ", lcmd)
    print("Actor's command: ", acmd)


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='Run instance script.')
    parser.add_argument('config_file', type=str, help='Config file name.')
    parser.add_argument('--dry-run', action='store_true')
    args = parser.parse_args()
    cfg = load_config(args.config_file)
    if args.dry_run:
        print(plan(cfg))
        print(*fab_commands(cfg, ['<actor-ip>'], '<learner-ip>'), sep='\n')
    else:
        run_instances_and_wait(cfg)
