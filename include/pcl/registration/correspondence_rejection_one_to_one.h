// pcl/registration/correspondence_rejection_one_to_one.h - see correspondence_rejection.h
#include <pcl/registration/correspondence_rejection.h>
